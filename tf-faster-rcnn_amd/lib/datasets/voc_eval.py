"""PASCAL VOC detection evaluation -- the step right after the device path.

Same entry points and results as the reference's lib/datasets/voc_eval.py (parse_rec :15-33, voc_ap :36-67,
voc_eval :70-214); host code, float64 like the reference.  Structure is this repo's own: annotations are cached in
binary mode (the reference opens its pickle cache in text mode, voc_eval.py:125, and cannot write it under Python 3),
detections are matched image by image with one overlap matrix per image instead of one Python iteration per detection.
The greedy rule is unchanged: detections in descending confidence; a detection is a true positive iff its best-overlap
ground truth has overlap > ovthresh, is not `difficult` and was not claimed before; a hit on a `difficult` box is
ignored; everything else is a false positive.
"""
import os
import pickle
import xml.etree.ElementTree as ET

import numpy as np


def parse_rec(filename):
    """One VOC annotation xml -> list of {'name','pose','truncated','difficult','bbox':[xmin,ymin,xmax,ymax]}."""
    out = []
    for node in ET.parse(filename).findall('object'):
        box = node.find('bndbox')
        out.append({'name': node.find('name').text,
                    'pose': node.find('pose').text,
                    'truncated': int(node.find('truncated').text),
                    'difficult': int(node.find('difficult').text),
                    'bbox': [int(box.find(k).text) for k in ('xmin', 'ymin', 'xmax', 'ymax')]})
    return out


def voc_ap(rec, prec, use_07_metric=False):
    """Average precision from a PR curve: VOC07 11-point interpolation, or the area under the precision envelope."""
    rec, prec = np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64)
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):                 # the reference's thresholds (incl. 0.30000000000000004)
            hit = rec >= t
            ap = ap + (np.max(prec[hit]) if hit.any() else 0) / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]           # precision envelope
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def load_annotations(annopath, imagenames, cachedir, imagesetfile):
    """{image name: parse_rec(...)} with a pickle cache next to the devkit (same file name as the reference's)."""
    if not os.path.isdir(cachedir):
        os.mkdir(cachedir)
    cachefile = os.path.join(cachedir, '%s_annots.pkl' % imagesetfile)
    if os.path.isfile(cachefile):
        with open(cachefile, 'rb') as f:
            try:
                return pickle.load(f)
            except Exception:
                f.seek(0)
                return pickle.load(f, encoding='bytes')
    recs = {name: parse_rec(annopath.format(name)) for name in imagenames}
    with open(cachefile, 'wb') as f:
        pickle.dump(recs, f)
    return recs


def _overlap_matrix(dets, gts):
    """VOC IoU (+1 pixel convention) of every detection [n,4] with every ground truth [g,4], float64."""
    ixmin = np.maximum(gts[None, :, 0], dets[:, None, 0])
    iymin = np.maximum(gts[None, :, 1], dets[:, None, 1])
    ixmax = np.minimum(gts[None, :, 2], dets[:, None, 2])
    iymax = np.minimum(gts[None, :, 3], dets[:, None, 3])
    iw = np.maximum(ixmax - ixmin + 1., 0.)
    ih = np.maximum(iymax - iymin + 1., 0.)
    inters = iw * ih
    uni = ((dets[:, None, 2] - dets[:, None, 0] + 1.) * (dets[:, None, 3] - dets[:, None, 1] + 1.) +
           (gts[None, :, 2] - gts[None, :, 0] + 1.) * (gts[None, :, 3] - gts[None, :, 1] + 1.) - inters)
    return inters / uni


def match_detections(image_ids, confidence, BB, class_recs, ovthresh):
    """tp / fp flags in descending-confidence order (voc_eval.py:160-203)."""
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    if nd == 0:
        return tp, fp
    order = np.argsort(-confidence)                          # the reference's call: tie order is numpy's
    ids = np.asarray(image_ids)[order]
    BB = np.asarray(BB, dtype=np.float64).reshape(nd, 4)[order]
    by_image = {}
    for pos, name in enumerate(ids):
        by_image.setdefault(name, []).append(pos)
    for name, positions in by_image.items():
        R = class_recs[name]
        gts = np.asarray(R['bbox'], dtype=np.float64)
        positions = np.asarray(positions)
        if gts.size == 0:
            fp[positions] = 1.
            continue
        ov = _overlap_matrix(BB[positions], gts.reshape(-1, 4))
        best, best_ov = ov.argmax(axis=1), ov.max(axis=1)
        claimed = R['det']
        for pos, j, o in zip(positions, best, best_ov):      # greedy claim in confidence order
            if o > ovthresh:
                if not R['difficult'][j]:
                    if not claimed[j]:
                        tp[pos] = 1.
                        claimed[j] = 1
                    else:
                        fp[pos] = 1.
            else:
                fp[pos] = 1.
    return tp, fp


def voc_eval(detpath, annopath, imagesetfile, classname, cachedir, ovthresh=0.5, use_07_metric=False, use_diff=False):
    """rec, prec, ap for one class.  detpath.format(classname): results file `image conf x1 y1 x2 y2` per line;
    annopath.format(image): annotation xml; imagesetfile: one image name per line."""
    with open(imagesetfile, 'r') as f:
        imagenames = [line.strip() for line in f.readlines()]
    recs = load_annotations(annopath, imagenames, cachedir, imagesetfile)

    class_recs, npos = {}, 0
    for name in imagenames:
        objs = [o for o in recs[name] if o['name'] == classname]
        difficult = np.zeros(len(objs), dtype=bool) if use_diff else np.array([o['difficult'] for o in objs]).astype(bool)
        npos += int(np.sum(~difficult))
        class_recs[name] = {'bbox': np.array([o['bbox'] for o in objs]), 'difficult': difficult, 'det': [False] * len(objs)}

    with open(detpath.format(classname), 'r') as f:
        rows = [line.strip().split(' ') for line in f.readlines()]
    image_ids = [r[0] for r in rows]
    confidence = np.array([float(r[1]) for r in rows])
    BB = np.array([[float(z) for z in r[2:]] for r in rows])

    tp, fp = match_detections(image_ids, confidence, BB, class_recs, ovthresh)
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)
