"""Detection result writers: the data formats on the output side of the path.

all_boxes[cls][image] = float array [n,5] (x1,y1,x2,y2,score) or [] -- what model.test.test_net returns and what the
reference's imdb.evaluate_detections consumes (lib/datasets/pascal_voc.py:203-219, lib/datasets/coco.py:258-292).
"""
import json
import os
import pickle

import numpy as np

from datasets.voc_eval import voc_eval


def _empty(dets):
    return isinstance(dets, list) and len(dets) == 0


def write_voc_results_file(all_boxes, classes, image_index, filename_template):
    """One `comp4_det_<set>_<cls>.txt` per foreground class; lines `image score x1 y1 x2 y2` with 1-based pixel
    coordinates (the VOCdevkit convention; pascal_voc.py:203-219).  Returns the files written."""
    written = []
    for c, cls in enumerate(classes):
        if cls == '__background__':
            continue
        path = filename_template.format(cls)
        with open(path, 'wt') as f:
            for i, index in enumerate(image_index):
                dets = all_boxes[c][i]
                if _empty(dets):
                    continue
                for d in np.asarray(dets):
                    f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(index, d[-1], d[0] + 1, d[1] + 1, d[2] + 1, d[3] + 1))
        written.append(path)
    return written


def do_python_eval(classes, filename_template, annopath, imagesetfile, cachedir, year, output_dir='output', use_diff=False,
                   verbose=True):
    """pascal_voc.py:221-263: AP per class (VOC07 11-point metric before 2010), mean AP, `<cls>_pr.pkl` files."""
    use_07_metric = int(year) < 2010
    if not os.path.isdir(output_dir):
        os.mkdir(output_dir)
    aps = []
    for cls in classes:
        if cls == '__background__':
            continue
        rec, prec, ap = voc_eval(filename_template.format(cls), annopath, imagesetfile, cls, cachedir, ovthresh=0.5,
                                 use_07_metric=use_07_metric, use_diff=use_diff)
        aps.append(ap)
        if verbose:
            print('AP for {} = {:.4f}'.format(cls, ap))
        with open(os.path.join(output_dir, cls + '_pr.pkl'), 'wb') as f:
            pickle.dump({'rec': rec, 'prec': prec, 'ap': ap}, f)
    if verbose:
        print('Mean AP = {:.4f}'.format(np.mean(aps)))
    return aps


def coco_results_one_category(boxes, image_index, cat_id):
    """coco.py:258-275: xyxy -> COCO xywh with the +1 width convention, one dict per detection."""
    results = []
    for i, index in enumerate(image_index):
        dets = boxes[i]
        if _empty(dets):
            continue
        dets = np.asarray(dets).astype(float)
        xs, ys = dets[:, 0], dets[:, 1]
        ws, hs = dets[:, 2] - xs + 1, dets[:, 3] - ys + 1
        results.extend({'image_id': index, 'category_id': cat_id, 'bbox': [xs[k], ys[k], ws[k], hs[k]], 'score': dets[k, -1]}
                       for k in range(dets.shape[0]))
    return results


def write_coco_results_file(all_boxes, classes, image_index, class_to_coco_cat_id, res_file):
    """coco.py:277-292: `[{"image_id", "category_id", "bbox": [x,y,w,h], "score"}, ...]` as json."""
    results = []
    for c, cls in enumerate(classes):
        if cls == '__background__':
            continue
        results.extend(coco_results_one_category(all_boxes[c], image_index, class_to_coco_cat_id[cls]))
    with open(res_file, 'w') as f:
        json.dump(results, f)
    return results
