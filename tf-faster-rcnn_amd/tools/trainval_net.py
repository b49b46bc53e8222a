#!/usr/bin/env python
"""Train a Faster R-CNN network on the MI355X path -- entry point of the reference's tools/trainval_net.py:29-139
with the same flags (--cfg --weight --imdb --imdbval --iters --tag --net --set).  Datasets / checkpoints are out
of scope: `--imdb synthetic` feeds seeded synthetic images + gt boxes.  Multi-GPU:
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/trainval_net.py ...` (one image per
rank per step, bucketed RCCL all-reduce of the gradients)."""
import argparse
import os
import pprint
import sys

import numpy as np
import torch

import _init_paths  # noqa: F401
from frcnn_hip import parallel
from frcnn_hip.runtime import Session
from model.config import cfg, cfg_from_file, cfg_from_list
from model.train_val import synthetic_data_layer, train_net
from nets.resnet_v1 import resnetv1


def parse_args():
    parser = argparse.ArgumentParser(description='Train a Faster R-CNN network')
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--weight', dest='weight', help='initialize with pretrained model weights (TF V2 checkpoint prefix, or .npz)', type=str)
    parser.add_argument('--output', dest='output_dir', help='directory for snapshots (default: none written)', default=None, type=str)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to train on', default='synthetic', type=str)
    parser.add_argument('--imdbval', dest='imdbval_name', help='dataset to validate on', default='synthetic', type=str)
    parser.add_argument('--iters', dest='max_iters', help='number of iterations to train', default=70000, type=int)
    parser.add_argument('--tag', dest='tag', help='tag of the model', default=None, type=str)
    parser.add_argument('--net', dest='net', help='res50, res101, res152', default='res50', type=str)
    parser.add_argument('--set', dest='set_cfgs', help='set config keys', default=None, nargs=argparse.REMAINDER)
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args()


if __name__ == '__main__':
    args = parse_args()
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    if args.set_cfgs is not None:
        cfg_from_list(args.set_cfgs)
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    all_reduce = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        all_reduce = parallel.make_grad_all_reduce()
    if rank == 0:
        print('Called with args:')
        print(args)
        print('Using config:')
        pprint.pprint(cfg)
    np.random.seed(cfg.RNG_SEED)
    if not args.net.startswith('res'):
        raise NotImplementedError('training is provided for the ResNet family (SURVEY.md 8a rows 14-17)')
    num_classes = 21
    sess = Session(seed=cfg.RNG_SEED)                                  # same weights on every rank
    net = resnetv1(num_layers=int(args.net[3:]))
    net.create_architecture("TRAIN", num_classes, tag='default', anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
    sess.init_variables(net.variable_specs())
    pretrained = None
    if args.weight and args.weight.endswith('.npz'):
        sess.load_variables(dict(np.load(args.weight)))
    elif args.weight:
        pretrained = args.weight                                        # ImageNet checkpoint: restore + fix_variables
    data = synthetic_data_layer(num_classes, seed=cfg.RNG_SEED + 1000 * rank, image_gain=1.0 / 256.0)
    out_dir = getattr(args, 'output_dir', None)
    # every rank resumes from the snapshots in out_dir (same weights, Momentum slots, iteration and sampling seed on all replicas);
    # only rank 0 writes new ones
    train_net(net, sess, data, max_iters=args.max_iters, all_reduce=all_reduce, world_size=world, pretrained_model=pretrained,
              output_dir=out_dir, write_snapshots=(rank == 0))
