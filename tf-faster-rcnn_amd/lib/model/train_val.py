"""model.train_val -- the training loop of the reference (lib/model/train_val.py:27-378) on the device chain.

SolverWrapper keeps the reference's schedule semantics: momentum SGD (MomentumOptimizer(lr, cfg.TRAIN.MOMENTUM),
:128), bias gradients doubled iff cfg.TRAIN.DOUBLE_BIAS (:133-143), L2 regularisation cfg.TRAIN.WEIGHT_DECAY,
learning rate multiplied by cfg.TRAIN.GAMMA at every cfg.TRAIN.STEPSIZE entry (:269-274), a progress line every
cfg.TRAIN.DISPLAY iterations in the reference's format (:298-302).  Data-parallel: one process per GPU, one image
per rank per step, one bucketed RCCL all-reduce of the flat gradient buffer (new relative to the reference, which
trains on a single GPU with IMS_PER_BATCH 1).  Checkpoints are TensorFlow V2 bundles read / written without TensorFlow
(frcnn_hip/tensor_bundle.py): `initialize` = ImageNet weights + fix_variables (:177-202), `snapshot` = Saver.save of the
variables, the Momentum slots and a `.pkl` with the iteration (:58-100), `restore` (:204-233).  roidb feeding / TensorBoard
are out of scope; `data_layer` is any iterator of blobs {'data','im_info','gt_boxes'}."""
import os
import pickle
import time

import numpy as np

from frcnn_hip.train import TrainState
from model.config import cfg


def find_previous(output_dir):
    """train_val.py:155-175: snapshots in output_dir, oldest first -> (count, pkl files, checkpoint prefixes).  A V2 bundle
    has no `.meta` graph file here, so `<prefix>.ckpt.index` marks a snapshot; the extra snapshots the reference takes right
    before a learning-rate step (iteration STEPSIZE + 1) are skipped like at :160-164."""
    import glob
    pattern = os.path.join(output_dir, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_*.ckpt.index')
    sfiles = sorted(glob.glob(pattern), key=os.path.getmtime)
    red = [os.path.join(output_dir, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_{:d}.ckpt.index'.format(st + 1)) for st in cfg.TRAIN.STEPSIZE]
    sfiles = [f[:-len('.index')] for f in sfiles if f not in red]
    nfiles = sorted(glob.glob(os.path.join(output_dir, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_*.pkl')), key=os.path.getmtime)
    red = [r.replace('.ckpt.index', '.pkl') for r in red]
    nfiles = [n for n in nfiles if n not in red]
    assert len(nfiles) == len(sfiles), 'snapshot .ckpt / .pkl files do not pair up in %s' % output_dir
    return len(sfiles), nfiles, sfiles


def remove_snapshot(np_paths, ss_paths):
    """train_val.py:235-256: keep the newest cfg.TRAIN.SNAPSHOT_KEPT snapshots, delete older ones (both files of the bundle)."""
    for _ in range(max(0, len(np_paths) - cfg.TRAIN.SNAPSHOT_KEPT)):
        nfile = str(np_paths.pop(0))
        if os.path.exists(nfile):                                 # another rank (or an earlier run) may have pruned it already
            os.remove(nfile)
    for _ in range(max(0, len(ss_paths) - cfg.TRAIN.SNAPSHOT_KEPT)):
        sfile = ss_paths.pop(0)
        for suffix in ('.data-00000-of-00001', '.index'):
            if os.path.exists(sfile + suffix):
                os.remove(sfile + suffix)


class SolverWrapper(object):
    def __init__(self, sess, network, data_layer, all_reduce=None, world_size=1, write_snapshots=True, force_dp=False):
        self.sess, self.net, self.data_layer = sess, network, data_layer
        self.write_snapshots = bool(write_snapshots)               # data-parallel runs: every rank READS snapshots, rank 0 writes them
        self.state = TrainState(sess, network, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WEIGHT_DECAY,
                                double_bias=cfg.TRAIN.DOUBLE_BIAS, bias_decay=cfg.TRAIN.BIAS_DECAY)
        self.state.all_reduce, self.state.world_size = all_reduce, world_size
        self.state.force_dp = bool(force_dp)                      # one replica under the data-parallel rules (TrainState.data_parallel)
        self.np_paths, self.ss_paths = [], []                     # snapshots written / found so far, oldest first

    # ---- checkpoints -------------------------------------------------------------------------------------------------
    def get_variables_in_checkpoint_file(self, file_name):
        """train_val.py:105-114 (pywrap_tensorflow.NewCheckpointReader(...).get_variable_to_shape_map())."""
        from frcnn_hip.tensor_bundle import open_checkpoint
        try:
            return open_checkpoint(file_name).get_variable_to_shape_map()
        except Exception as e:                                                    # the reference prints and carries on
            print(str(e))
            if "corrupted compressed block contents" in str(e):
                print("It's likely that your checkpoint file has been compressed with SNAPPY.")

    def initialize(self, pretrained_model):
        """train_val.py:177-202: restore what the ImageNet checkpoint holds, then the per-network fixes (RGB->BGR stem,
        VGG fc6/fc7 reshape, MobileNet input scale).  Call after sess.init_variables(net.variable_specs())."""
        print('Loading initial model weights from {:s}'.format(pretrained_model))
        var_keep_dic = self.get_variables_in_checkpoint_file(pretrained_model)
        names = self.net.get_variables_to_restore(list(self.sess.variables), var_keep_dic)
        self.sess.restore(pretrained_model, names)
        print('Loaded.')
        self.net.fix_variables(self.sess, pretrained_model)
        print('Fixed.')
        self.state.invalidate_prepared()
        return cfg.TRAIN.LEARNING_RATE, 0, list(cfg.TRAIN.STEPSIZE)

    def snapshot(self, it, output_dir):
        """train_val.py:58-100: `<prefix>_iter_<it>.ckpt` (variables + Momentum slots + global_step) and a .pkl with the
        iteration (the reference also pickles numpy RNG / data-layer cursors; the sampling here is seeded per step)."""
        if not self.write_snapshots:                             # data-parallel: rank 0 writes, the others only read
            return None, None
        os.makedirs(output_dir, exist_ok=True)
        base = os.path.join(output_dir, cfg.TRAIN.SNAPSHOT_PREFIX + '_iter_{:d}'.format(it))
        extra = self.state.export_variables(slots=True) if self.state.params else {}
        for k, v in extra.items():                                                # trained values replace the initial ones
            if not k.endswith("/Momentum"):
                self.sess.variables[k] = v
        extra["global_step"] = np.array(it, dtype=np.int64)
        self.sess.save(base + '.ckpt', {k: v for k, v in extra.items() if k.endswith("/Momentum") or k == "global_step"})
        with open(base + '.pkl', 'wb') as f:
            pickle.dump({'iter': it, 'sample_seed': int(self.net._sample_seed)}, f, pickle.HIGHEST_PROTOCOL)
        print('Wrote snapshot to: {:s}'.format(base + '.ckpt'))
        self.np_paths.append(base + '.pkl')
        self.ss_paths.append(base + '.ckpt')
        remove_snapshot(self.np_paths, self.ss_paths)             # keep cfg.TRAIN.SNAPSHOT_KEPT (train_val.py:309-311)
        return base + '.ckpt', base + '.pkl'

    def restore(self, sfile, nfile):
        """train_val.py:204-233: weights + optimizer slots + iteration from a snapshot; returns the iteration."""
        from frcnn_hip.tensor_bundle import BundleReader
        print('Restoring model snapshots from {:s}'.format(sfile))
        self.sess.restore(sfile)
        self.state.invalidate_prepared()
        self.state.pending_slots = BundleReader(sfile)           # imported right after TrainState.build(), before the first update
        with open(nfile, 'rb') as f:
            meta = pickle.load(f)
        self.net._sample_seed = meta.get('sample_seed', 0)        # the fg/bg sampling stream continues where it stopped
        if sfile not in self.ss_paths:
            self.np_paths.append(nfile)
            self.ss_paths.append(sfile)
        return meta['iter']

    def train_model(self, max_iters, verbose=True, start_iter=0, snapshot_dir=None):
        lr = cfg.TRAIN.LEARNING_RATE
        stepsizes = sorted(cfg.TRAIN.STEPSIZE, reverse=True)
        next_stepsize = stepsizes.pop() if stepsizes else None
        history = []
        t0 = time.time()
        while next_stepsize is not None and start_iter > next_stepsize:          # resumed past a step: :225-231
            lr *= cfg.TRAIN.GAMMA
            next_stepsize = stepsizes.pop() if stepsizes else None
        for it in range(start_iter + 1, max_iters + 1):
            if next_stepsize is not None and it == next_stepsize + 1:          # :269-274
                lr *= cfg.TRAIN.GAMMA
                next_stepsize = stepsizes.pop() if stepsizes else None
            self.state.lr = lr
            blobs = next(self.data_layer)
            # no host synchronisation inside a step: the losses stay on the device until somebody looks at them
            history.append(self.net.train_step_async(self.sess, blobs, self.state))
            if verbose and it % cfg.TRAIN.DISPLAY == 0:
                rpn_loss_cls, rpn_loss_box, loss_cls, loss_box, total_loss = history[-1].cpu().tolist()
                print('iter: %d / %d, total loss: %.6f\n >>> rpn_loss_cls: %.6f\n >>> rpn_loss_box: %.6f\n >>> loss_cls: %.6f\n'
                      ' >>> loss_box: %.6f\n >>> lr: %f' % (it, max_iters, total_loss, rpn_loss_cls, rpn_loss_box, loss_cls, loss_box, lr))
                print('speed: {:.3f}s / iter'.format((time.time() - t0) / (it - start_iter)))
            if snapshot_dir is not None and self.write_snapshots and it % cfg.TRAIN.SNAPSHOT_ITERS == 0:
                self.snapshot(it, snapshot_dir)
        self.enqueue_s = time.time() - t0          # the host's share: every step enqueued, the GPU not yet waited for (the read-back below waits)
        if history:
            import torch
            history = [float(v) for v in torch.stack(history)[:, 4].cpu().tolist()]
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


def train_net(network, sess, data_layer, max_iters=40000, all_reduce=None, world_size=1, pretrained_model=None, output_dir=None,
              resume=None, write_snapshots=True):
    """Train a Faster R-CNN network (reference signature minus imdb/roidb): pretrained_model = ImageNet checkpoint prefix
    (train_val.py:177-202), output_dir = where snapshots go every cfg.TRAIN.SNAPSHOT_ITERS, resume = (ckpt, pkl).
    Data-parallel runs pass the SAME output_dir to every rank (all replicas must resume from the same snapshot: weights,
    Momentum slots, iteration and sampling seed -- otherwise they diverge and issue different numbers of all-reduces) and
    write_snapshots = (rank == 0)."""
    sw = SolverWrapper(sess, network, data_layer, all_reduce=all_reduce, world_size=world_size, write_snapshots=write_snapshots)
    start = 0
    if resume is None and output_dir is not None and os.path.isdir(output_dir):
        lsf, nfiles, sfiles = find_previous(output_dir)            # train_val.py:243-252: continue from the newest snapshot
        if lsf:
            sw.np_paths, sw.ss_paths = list(nfiles[:-1]), list(sfiles[:-1])
            resume = (sfiles[-1], nfiles[-1])
    if resume is not None:
        start = sw.restore(*resume)
    elif pretrained_model is not None:
        sw.initialize(pretrained_model)
    print('Solving...')
    hist = sw.train_model(max_iters, start_iter=start, snapshot_dir=output_dir)
    if output_dir is not None and write_snapshots and max_iters % cfg.TRAIN.SNAPSHOT_ITERS:
        sw.snapshot(max_iters, output_dir)                        # the reference snapshots the last iteration too (:338-340)
    if world_size > 1 and output_dir is not None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()                                        # nobody returns (and resumes / evaluates) before rank 0's files are complete
    print('done solving')
    return hist
