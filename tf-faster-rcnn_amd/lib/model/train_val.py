"""model.train_val -- the training loop of the reference (lib/model/train_val.py:27-378) on the device chain.

SolverWrapper keeps the reference's schedule semantics: momentum SGD (MomentumOptimizer(lr, cfg.TRAIN.MOMENTUM),
:128), bias gradients doubled iff cfg.TRAIN.DOUBLE_BIAS (:133-143), L2 regularisation cfg.TRAIN.WEIGHT_DECAY,
learning rate multiplied by cfg.TRAIN.GAMMA at every cfg.TRAIN.STEPSIZE entry (:269-274), a progress line every
cfg.TRAIN.DISPLAY iterations in the reference's format (:298-302).  Data-parallel: one process per GPU, one image
per rank per step, one bucketed RCCL all-reduce of the flat gradient buffer (new relative to the reference, which
trains on a single GPU with IMS_PER_BATCH 1).  Checkpointing / roidb feeding / TensorBoard are out of scope
(SURVEY.md 2); `data_layer` is any iterator of blobs {'data','im_info','gt_boxes'}."""
import time

import numpy as np

from frcnn_hip.train import TrainState
from model.config import cfg


class SolverWrapper(object):
    def __init__(self, sess, network, data_layer, all_reduce=None, world_size=1):
        self.sess, self.net, self.data_layer = sess, network, data_layer
        self.state = TrainState(sess, network, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WEIGHT_DECAY,
                                double_bias=cfg.TRAIN.DOUBLE_BIAS, bias_decay=cfg.TRAIN.BIAS_DECAY)
        self.state.all_reduce, self.state.world_size = all_reduce, world_size

    def train_model(self, max_iters, verbose=True):
        lr = cfg.TRAIN.LEARNING_RATE
        stepsizes = sorted(cfg.TRAIN.STEPSIZE, reverse=True)
        next_stepsize = stepsizes.pop() if stepsizes else None
        history = []
        t0 = time.time()
        for it in range(1, max_iters + 1):
            if next_stepsize is not None and it == next_stepsize + 1:          # :269-274
                lr *= cfg.TRAIN.GAMMA
                next_stepsize = stepsizes.pop() if stepsizes else None
            self.state.lr = lr
            blobs = next(self.data_layer)
            rpn_loss_cls, rpn_loss_box, loss_cls, loss_box, total_loss = self.net.train_step(self.sess, blobs, self.state)
            history.append(total_loss)
            if verbose and it % cfg.TRAIN.DISPLAY == 0:
                print('iter: %d / %d, total loss: %.6f\n >>> rpn_loss_cls: %.6f\n >>> rpn_loss_box: %.6f\n >>> loss_cls: %.6f\n'
                      ' >>> loss_box: %.6f\n >>> lr: %f' % (it, max_iters, total_loss, rpn_loss_cls, rpn_loss_box, loss_cls, loss_box, lr))
                print('speed: {:.3f}s / iter'.format((time.time() - t0) / it))
        return history


def synthetic_data_layer(num_classes, seed=3, height=600, width=1000, scale=1.6, image_gain=1.0):
    """Seeded stand-in for lib/roi_data_layer/layer.py (SURVEY.md 8d 'Training gt'): 3-20 boxes per image."""
    rng = np.random.RandomState(seed)
    while True:
        n = rng.randint(3, 21)
        w, h = 32 + rng.rand(n) * 368, 32 + rng.rand(n) * 368
        x1 = rng.rand(n) * np.maximum(width - w, 1)
        y1 = rng.rand(n) * np.maximum(height - h, 1)
        gt = np.stack([x1, y1, np.minimum(x1 + w, width - 1), np.minimum(y1 + h, height - 1), rng.randint(1, num_classes, size=n)],
                      axis=1).astype(np.float32)
        image = ((rng.rand(1, height, width, 3) * 255.0).astype(np.float32) - cfg.PIXEL_MEANS.astype(np.float32)) * np.float32(image_gain)
        yield {'data': image, 'im_info': np.array([height, width, scale], dtype=np.float32), 'gt_boxes': gt}


def train_net(network, sess, data_layer, max_iters=40000, all_reduce=None, world_size=1):
    """Train a Faster R-CNN network (reference signature minus imdb/roidb/output dirs)."""
    sw = SolverWrapper(sess, network, data_layer, all_reduce=all_reduce, world_size=world_size)
    print('Solving...')
    hist = sw.train_model(max_iters)
    print('done solving')
    return hist
