#!/usr/bin/env python
"""Test a Faster R-CNN network on the MI355X path -- the entry point of the reference's
tools/test_net.py:27-122 with the same flags (--cfg --model --imdb --comp --num_dets --tag --net --set).

Datasets, checkpoints and cv2 are out of scope: `--imdb synthetic_N` runs N seeded
600x1000 synthetic images through the device chain; `--model file.npz` loads variables stored under
their TF/slim names, otherwise the reference initialisers are used."""
import argparse
import pprint
import sys
import time

import numpy as np

import _init_paths  # noqa: F401
from frcnn_hip.runtime import Session
from model.config import cfg, cfg_from_file, cfg_from_list
from model.test import test_net
from nets.mobilenet_v1 import mobilenetv1
from nets.resnet_v1 import resnetv1
from nets.vgg16 import vgg16


def parse_args():
    parser = argparse.ArgumentParser(description='Test a Faster R-CNN network')
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--model', dest='model', help='model to test (.npz of TF-named variables)', default=None, type=str)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to test', default='synthetic_8', type=str)
    parser.add_argument('--comp', dest='comp_mode', help='competition mode', action='store_true')
    parser.add_argument('--num_dets', dest='max_per_image', help='max number of detections per image', default=100, type=int)
    parser.add_argument('--tag', dest='tag', help='tag of the model', default='', type=str)
    parser.add_argument('--net', dest='net', help='vgg16, res50, res101, res152, mobile', default='res50', type=str)
    parser.add_argument('--set', dest='set_cfgs', help='set config keys', default=None, nargs=argparse.REMAINDER)
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args()


if __name__ == '__main__':
    args = parse_args()
    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs)
    print('Using config:')
    pprint.pprint(cfg)

    if not args.imdb_name.startswith('synthetic'):
        raise SystemExit('only --imdb synthetic_N is available: dataset readers are out of scope (SURVEY.md section 2)')
    n_images = int(args.imdb_name.split('_')[1]) if '_' in args.imdb_name else 8
    num_classes = 21
    tag = args.tag if args.tag else 'default'

    sess = Session(seed=cfg.RNG_SEED)
    if args.net == 'vgg16':
        net = vgg16()
    elif args.net in ('res50', 'res101', 'res152'):
        net = resnetv1(num_layers=int(args.net[3:]))
    elif args.net == 'mobile':
        net = mobilenetv1()
    else:
        raise NotImplementedError
    net.create_architecture("TEST", num_classes, tag=tag, anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    if args.model:
        print('Loading model check point from {:s}'.format(args.model))
        sess.init_variables(net.variable_specs())
        sess.load_variables(dict(np.load(args.model)))
        print('Loaded.')
    else:
        print('No --model given: initialising with the reference initialisers (seed %d)' % cfg.RNG_SEED)
        sess.init_variables(net.variable_specs())

    rng = np.random.RandomState(cfg.RNG_SEED)
    scale = 1.6
    images = [((rng.rand(1, 600, 1000, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32), scale, (375, 625))
              for _ in range(n_images)]
    t0 = time.time()
    all_boxes = test_net(sess, net, images, max_per_image=args.max_per_image)
    print('%d images in %.3fs; detections per image: %s' %
          (n_images, time.time() - t0, [sum(len(all_boxes[j][i]) for j in range(1, num_classes)) for i in range(n_images)]))
