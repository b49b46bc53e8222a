#!/usr/bin/env python
"""Inference entry point on the MI355X path.

Mirrors the command line of the reference's tools/test_net.py (flags --cfg --model --imdb --comp --num_dets --tag --net
--set, /root/reference/tools/test_net.py:27-49) so existing launch scripts keep working.  What differs: there is no
TensorFlow session -- `--model` takes a TensorFlow V2 checkpoint prefix (read without TensorFlow) or an .npz of variables
under their TF/slim names (otherwise the reference initialisers are used); `--imdb synthetic_N` pushes N seeded 600x1000
images through the device chain, `--imdb voc_2007_test` reads `<cfg.DATA_DIR>/VOCdevkit2007` (JPEGs via PIL), runs the
raw-image device path and the VOC evaluation (datasets/pascal_voc.py); other dataset readers are out of scope."""
import argparse
import pprint
import sys
import time

import numpy as np

import _init_paths  # noqa: F401  (adds tf-faster-rcnn_amd/ and tf-faster-rcnn_amd/lib to sys.path)
from frcnn_hip.runtime import Session
from model.config import cfg, cfg_from_file, cfg_from_list
from model.test import test_net
from nets.mobilenet_v1 import mobilenetv1
from nets.resnet_v1 import resnetv1
from nets.vgg16 import vgg16

FLAGS = [  # (flag, dest, kwargs) -- same names / dests as the reference parser
    ("--cfg", "cfg_file", dict(type=str, default=None, help="optional config file")),
    ("--model", "model", dict(type=str, default=None, help="TF V2 checkpoint prefix, or .npz of variables (TF/slim names)")),
    ("--imdb", "imdb_name", dict(type=str, default="synthetic_8", help="synthetic_N | voc_<year>_<split>")),
    ("--comp", "comp_mode", dict(action="store_true", help="competition mode (accepted, unused)")),
    ("--num_dets", "max_per_image", dict(type=int, default=100, help="max number of detections per image")),
    ("--tag", "tag", dict(type=str, default="", help="tag of the model")),
    ("--net", "net", dict(type=str, default="res50", help="vgg16, res50, res101, res152, mobile")),
    ("--set", "set_cfgs", dict(nargs=argparse.REMAINDER, default=None, help="set config keys")),
]
NETS = {"vgg16": vgg16, "mobile": mobilenetv1, "res50": lambda: resnetv1(num_layers=50),
        "res101": lambda: resnetv1(num_layers=101), "res152": lambda: resnetv1(num_layers=152)}


def build_parser():
    ap = argparse.ArgumentParser(description="Test a Faster R-CNN network (MI355X path)")
    for flag, dest, kw in FLAGS:
        ap.add_argument(flag, dest=dest, **kw)
    return ap


def synthetic_images(n, scale=1.6, height=600, width=1000):
    rng = np.random.RandomState(cfg.RNG_SEED)
    means = cfg.PIXEL_MEANS.astype(np.float32)
    orig = (int(height / scale), int(width / scale))
    for _ in range(n):
        yield (rng.rand(1, height, width, 3) * 255.0).astype(np.float32) - means, scale, orig


def main(argv):
    ap = build_parser()
    if not argv:
        ap.print_help()
        return 1
    args = ap.parse_args(argv)
    if args.cfg_file:
        cfg_from_file(args.cfg_file)
    if args.set_cfgs:
        cfg_from_list(args.set_cfgs)
    print("Called with args:\n%s\nUsing config:" % (args,))
    pprint.pprint(cfg)
    imdb = None
    if args.imdb_name.startswith("voc_"):
        import os
        from datasets.pascal_voc import pascal_voc
        _, year, split = args.imdb_name.split("_")
        imdb = pascal_voc(split, year, os.path.join(cfg.DATA_DIR, "VOCdevkit" + year))
        imdb.competition_mode(args.comp_mode)
    elif not args.imdb_name.startswith("synthetic"):
        raise SystemExit("--imdb synthetic_N or voc_<year>_<split>: other dataset readers are out of scope (SURVEY.md section 2)")
    if args.net not in NETS:
        raise NotImplementedError(args.net)
    n_images = int(args.imdb_name.split("_")[1]) if (imdb is None and "_" in args.imdb_name) else 8
    num_classes = 21
    net = NETS[args.net]()
    net.create_architecture("TEST", num_classes, tag=args.tag or "default", anchor_scales=cfg.ANCHOR_SCALES,
                            anchor_ratios=cfg.ANCHOR_RATIOS)
    sess = Session(seed=cfg.RNG_SEED)
    sess.init_variables(net.variable_specs())
    if args.model and args.model.endswith(".npz"):
        print("Loading variables from %s" % args.model)
        sess.load_variables(dict(np.load(args.model)))
    elif args.model:
        print("Loading model check point from {:s}".format(args.model))          # tools/test_net.py:110-114
        sess.restore(args.model)
        print("Loaded.")
    else:
        print("No --model: reference initialisers, seed %d" % cfg.RNG_SEED)
    if imdb is not None:
        import os
        from model.test import test_net_imdb
        out_dir = os.path.join(cfg.ROOT_DIR, "output", args.net, imdb.name, args.tag or "default")
        test_net_imdb(sess, net, imdb, out_dir, max_per_image=args.max_per_image)
        return 0
    t0 = time.time()
    all_boxes = test_net(sess, net, synthetic_images(n_images), max_per_image=args.max_per_image)
    per_image = [sum(len(all_boxes[j][i]) for j in range(1, num_classes)) for i in range(n_images)]
    print("%d images in %.3fs; detections per image: %s" % (n_images, time.time() - t0, per_image))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
