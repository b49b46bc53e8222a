#!/usr/bin/env python
"""Re-evaluate a finished test run: `reval.py <output_dir> [--imdb voc_2007_test] [--comp] [--nms]`.

Same command line as the reference's tools/reval.py:22-46 (`--matlab` is accepted and ignored: there is no MATLAB eval
here): loads `<output_dir>/detections.pkl`, optionally applies NMS at cfg.TEST.NMS to every class/image list
(model.test.apply_nms, on the device NMS), writes the VOC results files and prints the AP table."""
import argparse
import os
import pickle
import sys

import _init_paths  # noqa: F401
from model.config import cfg, cfg_from_list
from model.test import apply_nms


def parse_args(argv):
    parser = argparse.ArgumentParser(description='Re-evaluate results')
    parser.add_argument('output_dir', nargs=1, help='results directory', type=str)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to re-evaluate', default='voc_2007_test', type=str)
    parser.add_argument('--matlab', dest='matlab_eval', help='use matlab for evaluation (ignored)', action='store_true')
    parser.add_argument('--comp', dest='comp_mode', help='competition mode', action='store_true')
    parser.add_argument('--nms', dest='apply_nms', help='apply nms', action='store_true')
    parser.add_argument('--set', dest='set_cfgs', help='set config keys', default=None, nargs=argparse.REMAINDER)
    if not argv:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args(argv)


def from_dets(imdb_name, output_dir, args):
    from datasets.pascal_voc import pascal_voc
    if not imdb_name.startswith('voc_'):
        raise SystemExit('--imdb voc_<year>_<split>: other dataset readers are out of scope (SURVEY.md section 2)')
    _, year, split = imdb_name.split('_')
    imdb = pascal_voc(split, year, os.path.join(cfg.DATA_DIR, 'VOCdevkit' + year))
    imdb.competition_mode(args.comp_mode)
    with open(os.path.join(output_dir, 'detections.pkl'), 'rb') as f:
        dets = pickle.load(f)
    if args.apply_nms:
        print('Applying NMS to all detections')
        dets = apply_nms(dets, cfg.TEST.NMS)
    print('Evaluating detections')
    return imdb.evaluate_detections(dets, output_dir)


def main(argv):
    args = parse_args(argv)
    if args.set_cfgs:
        cfg_from_list(args.set_cfgs)
    from_dets(args.imdb_name, os.path.abspath(args.output_dir[0]), args)
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
