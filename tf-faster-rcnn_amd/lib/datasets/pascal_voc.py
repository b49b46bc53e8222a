"""Evaluation-side view of a PASCAL VOC devkit: image index, class list, results files, AP.

Covers what `imdb.evaluate_detections(all_boxes, output_dir)` needs at the end of model.test.test_net
(reference: lib/datasets/pascal_voc.py:27-50 constructor fields, :92-103 image index, :186-201 results path,
:203-263 writer + python eval, :281-296 evaluate_detections / competition_mode).  The training-side roidb machinery
(gt_roidb, flipping, selective search) is data-loader plumbing outside the device path and is not provided.
"""
import os
import uuid

from datasets import results

VOC_CLASSES = ('__background__', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


class pascal_voc(object):
    def __init__(self, image_set, year, devkit_path, use_diff=False, classes=VOC_CLASSES):
        self.name = 'voc_' + year + '_' + image_set + ('_diff' if use_diff else '')
        self._year, self._image_set, self._devkit_path = year, image_set, devkit_path
        self._data_path = os.path.join(devkit_path, 'VOC' + year)
        self._classes = tuple(classes)
        self._salt = str(uuid.uuid4())
        self._comp_id = 'comp4'
        self.config = {'cleanup': True, 'use_salt': True, 'use_diff': use_diff}
        if not os.path.exists(self._data_path):
            raise IOError('Path does not exist: {}'.format(self._data_path))
        with open(self._image_set_file()) as f:
            self._image_index = [line.strip() for line in f.readlines()]

    classes = property(lambda self: self._classes)
    num_classes = property(lambda self: len(self._classes))
    image_index = property(lambda self: self._image_index)
    num_images = property(lambda self: len(self._image_index))

    def _image_set_file(self):
        return os.path.join(self._data_path, 'ImageSets', 'Main', self._image_set + '.txt')

    def image_path_at(self, i):
        return os.path.join(self._data_path, 'JPEGImages', self._image_index[i] + '.jpg')

    def _get_comp_id(self):
        return self._comp_id + '_' + self._salt if self.config['use_salt'] else self._comp_id

    def _get_voc_results_file_template(self):
        # VOCdevkit/results/VOC2007/Main/<comp_id>_det_test_aeroplane.txt
        d = os.path.join(self._devkit_path, 'results', 'VOC' + self._year, 'Main')
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, self._get_comp_id() + '_det_' + self._image_set + '_{:s}.txt')

    def evaluate_detections(self, all_boxes, output_dir, verbose=True):
        template = self._get_voc_results_file_template()
        files = results.write_voc_results_file(all_boxes, self._classes, self._image_index, template)
        aps = results.do_python_eval(self._classes, template, os.path.join(self._data_path, 'Annotations', '{:s}.xml'),
                                     self._image_set_file(), os.path.join(self._devkit_path, 'annotations_cache'), self._year,
                                     output_dir, self.config['use_diff'], verbose)
        if self.config['cleanup']:
            for path in files:
                os.remove(path)
        return aps

    def competition_mode(self, on):
        self.config['use_salt'] = not on
        self.config['cleanup'] = not on
