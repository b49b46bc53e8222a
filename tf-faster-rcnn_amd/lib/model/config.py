"""cfg -- the reference's global configuration object, re-hosted without easydict/TensorFlow.

Same keys, defaults and helper functions as /root/reference/lib/model/config.py:14-387
(cfg, cfg_from_file, cfg_from_list, get_output_dir, get_output_tb_dir) so that the tools/ entry
points and `--set KEY VALUE` overrides behave identically.  Differences, on purpose:
  * yaml.safe_load instead of the Loader-less yaml.load (config.py:362 raises under PyYAML >= 6);
  * USE_GPU_NMS (default True, as in the reference) keeps its meaning "the CUDA kernel's NMS": the HIP kernels then apply
    that kernel's rule, suppress iff IoU > thresh in float32 (lib/nms/nms_kernel.cu:71); False selects the Cython rule,
    suppress iff (double)IoU >= thresh (lib/nms/cpu_nms.pyx:65) -- the path BASELINE.json pins (bench.py and the parity
    tests set it).  Both run on the GPU; the two rules differ only when an IoU hits the threshold exactly;
  * USE_E2E_TF defaults to False (the reference: True): the numpy/Cython layer semantics are the path BASELINE.json
    names.  True selects the reference's TF-op semantics on the same kernels: int32-truncated anchors, proposal_layer_tf
    (tf.image.non_max_suppression over ALL anchors: no +1 areas, `>`), proposal_top_layer_tf.
"""
import ast
import os
import os.path as osp

import numpy as np


class AttrDict(dict):
    """dict with attribute access, recursive on assignment (what the reference uses easydict for)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


_TRAIN = dict(
    LEARNING_RATE=0.001, MOMENTUM=0.9, WEIGHT_DECAY=0.0001, GAMMA=0.1, STEPSIZE=[30000], DISPLAY=10,
    DOUBLE_BIAS=True, TRUNCATED=False, BIAS_DECAY=False, USE_GT=False, ASPECT_GROUPING=False,
    SNAPSHOT_KEPT=3, SUMMARY_INTERVAL=180, SCALES=(600,), MAX_SIZE=1000, IMS_PER_BATCH=1, BATCH_SIZE=128,
    FG_FRACTION=0.25, FG_THRESH=0.5, BG_THRESH_HI=0.5, BG_THRESH_LO=0.1, USE_FLIPPED=True, BBOX_REG=True,
    BBOX_THRESH=0.5, SNAPSHOT_ITERS=5000, SNAPSHOT_PREFIX='res101_faster_rcnn', BBOX_NORMALIZE_TARGETS=True,
    BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0), BBOX_NORMALIZE_TARGETS_PRECOMPUTED=True,
    BBOX_NORMALIZE_MEANS=(0.0, 0.0, 0.0, 0.0), BBOX_NORMALIZE_STDS=(0.1, 0.1, 0.2, 0.2), PROPOSAL_METHOD='gt',
    HAS_RPN=True, RPN_POSITIVE_OVERLAP=0.7, RPN_NEGATIVE_OVERLAP=0.3, RPN_CLOBBER_POSITIVES=False,
    RPN_FG_FRACTION=0.5, RPN_BATCHSIZE=256, RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=12000,
    RPN_POST_NMS_TOP_N=2000, RPN_BBOX_INSIDE_WEIGHTS=(1.0, 1.0, 1.0, 1.0), RPN_POSITIVE_WEIGHT=-1.0,
    USE_ALL_GT=True)

_TEST = dict(
    SCALES=(600,), MAX_SIZE=1000, NMS=0.3, SVM=False, BBOX_REG=True, HAS_RPN=False, PROPOSAL_METHOD='gt',
    RPN_NMS_THRESH=0.7, RPN_PRE_NMS_TOP_N=6000, RPN_POST_NMS_TOP_N=300, MODE='nms', RPN_TOP_N=5000)

__C = AttrDict(
    TRAIN=_TRAIN, TEST=_TEST,
    RESNET=dict(MAX_POOL=False, FIXED_BLOCKS=1),
    MOBILENET=dict(REGU_DEPTH=False, FIXED_LAYERS=5, WEIGHT_DECAY=0.00004, DEPTH_MULTIPLIER=1.),
    PIXEL_MEANS=np.array([[[102.9801, 115.9465, 122.7717]]]), RNG_SEED=3,
    ROOT_DIR=osp.abspath(osp.join(osp.dirname(__file__), '..', '..')),
    MATLAB='matlab', EXP_DIR='default', USE_GPU_NMS=True, USE_E2E_TF=False, POOLING_MODE='crop', POOLING_SIZE=7,
    ANCHOR_SCALES=[8, 16, 32], ANCHOR_RATIOS=[0.5, 1, 2], RPN_CHANNELS=512,
    # device-path switches (no reference counterpart): Winograd F(m x m,3x3) for the 3x3 stride-1 convolutions at test
    # time; m = WINOGRAD_M (4 or 2) except scopes containing a WINOGRAD_F2_SCOPES token, which use m = 2.  WINOGRAD_TRAIN: also in
    # the training step (forward and data gradient of those layers; filters transformed on the device every step).
    # Shipped policy (profiles/r02_fullsize_parity.txt): F(2x2,3x3) in block1 / block2 -- the early layers, whose activations carry a
    # large common mean, are where the F(4x4,3x3) transforms lose digits (full-size head error 1.8x the float32 control with F(4,3)
    # everywhere, 1.0x with this policy, for 1.2 % of throughput); F(4x4,3x3) in block3 / RPN / block4 (7x7 scheme).
    # WINOGRAD_DIRECT_SCOPES: scopes containing one of these tokens keep the direct implicit-GEMM kernel.
    # FUSE_TAIL_MEAN: TEST mode, the tail's last 1x1 convolution + reduce_mean in ONE launch (frcnn_gemm_h2_mean: the [R*49, 2048] tensor is
    # neither written nor re-read: 964 MB less per 4-image batch, one launch less; +0.3 % images/s, profiles/r04_q_ab_fused_mean.txt) where
    # that convolution runs in frcnn_gemm_h2.  One batch entry per image: the 49 rows of a RoI are added in an order that depends on the
    # RoI's index inside its image only, so the same image gives bit-identical fc7 in every batch slot and at every batch size (round 3's
    # f32 version added them in tile order, i.e. slot-dependent, and was off for that reason; it is gone).
    # WINOGRAD_7X7: 7x7 maps (per-RoI crops) use the mixed F(4,3)+F(3,3) scheme (121 instead of 144 products per RoI)
    # MFMA_X3: TEST mode, the large pointwise convolutions and Winograd products (plain GEMMs with Cout % 64 == 0, Cin % 32 == 0, >= 150 tiles) run on the bf16
    # matrix pipe with EXACTLY split f32 operands (csrc/gemm_x3.hip: x = h + m + l, six bf16 MFMAs per f32 product, f32 accumulation,
    # dropped terms <= 2^-24 relative): 1.3-1.6x faster than the f32 MFMA and, measured against float64, slightly MORE accurate than
    # it (profiles/r02_m_x3_sweep.txt).  False = every product on v_mfma_f32_32x32x2_f32 (bench.py --mfma f32).
    # MFMA_H2: TEST mode, plain GEMMs with Cin % 128 == 0 and Cout % 128 == 0 run on the fp16 matrix pipe with BLOCK-SCALED two-piece
    # operands (csrc/gemm_h2.hip: x = 2^-e (h + l) per 128-k block, three fp16 MFMAs per f32 product, f32 accumulation, dropped terms
    # <= 3 * 2^-22 relative, measured 1e-7 of the output scale -- below the f32 MFMA kernel's own error): 1.3-1.7x faster than MFMA_X3.
    # The producers (GEMM epilogue, Winograd transforms) emit the next layer's operand planes; H2_LAZY_SPLIT: an eligible layer
    # whose input has no planes yet splits it with a separate pass (False: such a layer takes the X3 / f32 path instead).
    # H2_TRUNK_PLANES: inside a run of h2 bottleneck units with identity shortcuts the unit output (the residual trunk) is kept as operand
    # planes ONLY -- the next unit's conv1 reads them as its input and its conv3 reads them as the residual ((h + l) * 2^-e, >= 22
    # significant bits: the stored trunk rounds at 2^-23 relative instead of 2^-24); the float32 tensor is written only where a
    # non-GEMM consumer follows (block ends, RPN / crop, the spatial mean).  Saves a third of conv3's HBM traffic.  Rounds 3 / 4 measured
    # +0.6 % / -0.2 % images/s (the epilogue was bound by its serial tile boundary, not by bytes) and kept it off; with round 5's light tile
    # boundary the bytes show: +3.0 % (515.1 / 516.4 -> 533.0 / 529.7 images/s, interleaved on one box, profiles/r05_f_ab_trunk_planes.txt),
    # every config of the full-size harness inside the float32 control's own loss (policy `shipped`; `shipped_f32trunk` is the old form): ON.
    # H2_TRAIN: TRAIN mode too -- the pointwise convolutions of the forward pass and their data gradients (>= H2_MIN_TILES tiles, i.e. the
    # RoI tail) run in frcnn_gemm_h2; filters are re-split after every solver step, float32 activations are kept for the tape.
    # H2_TRAIN_MIN_TILES: the TRAIN-mode threshold (tiles of 128 x 128 a launch must have); None = TRAIN mode reads H2_MIN_TILES like TEST
    # mode (every value means what it says -- nothing is compared against a default).  One image per step: 150 puts block3's conv3
    # (152 tiles) and block2's conv3 (300) on frcnn_gemm_h2, where the split-K f32 kernel is the shorter launch at that size -- 320 keeps only
    # the RoI tail (392 ... 1 568 tiles): 17.17 -> 16.93 ms per step (160: 17.07; 1000: 18.0; 38: 18.2; profiles/r05_w_*), set to 320.
    # WGRAD_STREAM: the reverse sweep enqueues the filter gradients (operand transposes, split-K GEMM, bias column sum) round-robin on this
    # many side HIP streams beside the data-gradient chain (0: all on one stream); joined before the solver.  Same kernels, same bits --
    # only the overlap changes.  Data-parallel runs use at most one (the bucketed all-reduce orders itself after a single stream).
    # WGRAD_TN: filter gradients of convolutions with Cin, Cout % 64 == 0 by frcnn_conv2d_wgrad (dW = dY^T X read from the NHWC tensors as
    # they lie, csrc/wgrad_tn.hip) instead of transpose_pad / im2col_t + the forward GEMM kernel.
    # WGRAD_H2 (with MFMA_H2 and H2_TRAIN): those gradients on the fp16 matrix pipe, operands split in registers (csrc/wgrad_h2.hip).
    # PREP_STREAM: the weight-only launches of the data-gradient chain (flipped / transposed filters, their h2 split, Winograd transforms of
    # the gradient filters) are re-run by the solver right after the update, on their own stream beside the next forward pass.
    # X3_TILE_CFG: frcnn_gemm_x3's per-call tile configuration (-1 = by shape; A/B runs).
    # Round-3 switches that measured no gain were removed in round 4 (OVERLAP_TAIL_ENTRY, TRAIN_GRAPH, H2_TRAIN_WINO, X3_TERMS = 9): the
    # code is kept as scratch/r04_pruned_switches.patch (git apply -R restores it), the measurements in profiles/r02_c_sweep.txt,
    # r03_ab_c5_streams_graph.txt, r02_r_x3_9terms.txt.
    # TRAIN_REPLAY: the training step as ONE recorded launch list (frcnn_hip/replay.py; the reference's step is one sess.run of a graph
    # built once, network.py:488-498): the second step of an image shape is recorded while it runs eagerly, later steps of that shape
    # replay the list (same launches, arguments, streams, order -> same bits) without the Python a step costs the host (~14 ms for ~1 300
    # launches).  False: every step is enqueued by the Python code (the form rounds 1-4 measured).
    # TRAIN_PICK_STREAMS (with TRAIN_REPLAY): n > 0 -- the physical streams of the step's helper slots (filter gradients, solver, filter
    # preparation) are chosen once per session by timing real steps over a pool of n fresh streams (frcnn_hip/replay.py StreamPicker): which
    # hardware queue a stream lands on depends on who created streams before (an RCCL group that merely exists: +4 ms per step in round 4;
    # with the picker the data-parallel-rules step went from 25.2 to 20.0 ms, profiles/r05_j_c5_dp_rules_ab_stream_picker.txt).  Default 6:
    # 1 + 6 x (helper slots) windows of 3 real steps at start-up (~2 s); 0 = keep the streams the step was recorded on.
    # H2_TILE_CFG: -1 = tile shape by launch size (csrc/gemm_h2.hip), else a frcnn_gemm_h2 configuration id for every launch (A/B runs).
    # GRAPH_CACHE_SHAPES / TRAIN_CACHE_SHAPES: image shapes per network tag whose captured hipGraph (TEST) / recorded steps (TRAIN) and static
    # buffers are kept; the least recently used shape is dropped first and its buffers go back to the allocator (Session.shape_scope).  The
    # reference's graph takes any [1, H, W, 3] (lib/nets/network.py:386-390) and an imdb has hundreds of sizes: memory is bounded by these
    # caps, not by the imdb.  ResNet-101 at 600 x 1000 holds ~2 GB per shape.
    HIP=dict(WINOGRAD=True, WINOGRAD_MIN_CIN=64, WINOGRAD_M=4, WINOGRAD_F2_SCOPES=("block1", "block2"), WINOGRAD_DIRECT_SCOPES=(),
             WINOGRAD_TRAIN=True,
             WINOGRAD_7X7=True, FUSE_TAIL_MEAN=True, MFMA_X3=True,
             MFMA_H2=True, H2_LAZY_SPLIT=True, H2_MIN_TILES=150, H2_TRAIN_MIN_TILES=320, H2_TRUNK_PLANES=True, H2_TILE_CFG=-1,
             X3_TILE_CFG=-1, H2_TRAIN=True, WGRAD_STREAM=2, WGRAD_TN=True, WGRAD_H2=True, PREP_STREAM=True, TRAIN_REPLAY=True, TRAIN_PICK_STREAMS=6,
             GRAPH_CACHE_SHAPES=4, TRAIN_CACHE_SHAPES=16))
__C.DATA_DIR = osp.abspath(osp.join(__C.ROOT_DIR, 'data'))
cfg = __C


def get_output_dir(imdb, weights_filename):
    """output/<EXP_DIR>/<imdb.name>[/<weights_filename>]  (reference config.py:293-306)."""
    outdir = osp.abspath(osp.join(__C.ROOT_DIR, 'output', __C.EXP_DIR, imdb.name))
    outdir = osp.join(outdir, 'default' if weights_filename is None else weights_filename)
    os.makedirs(outdir, exist_ok=True)
    return outdir


def get_output_tb_dir(imdb, weights_filename):
    outdir = osp.abspath(osp.join(__C.ROOT_DIR, 'tensorboard', __C.EXP_DIR, imdb.name))
    outdir = osp.join(outdir, 'default' if weights_filename is None else weights_filename)
    os.makedirs(outdir, exist_ok=True)
    return outdir


_NULLABLE_KEYS = ("H2_TRAIN_MIN_TILES",)      # int or None (None = "the same as H2_MIN_TILES"): the two loaders accept either


def _merge_a_into_b(a, b):
    """Recursive merge with the reference's type checks (config.py:325-356)."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        old = b[k]
        if type(old) is not type(v) and not (isinstance(old, dict) and isinstance(v, dict)):
            if isinstance(old, np.ndarray):
                v = np.array(v, dtype=old.dtype)
            elif isinstance(old, tuple) and isinstance(v, list):
                v = tuple(v)
            elif isinstance(old, list) and isinstance(v, tuple):
                v = list(v)
            elif k in _NULLABLE_KEYS and (v is None or old is None) and isinstance(old if v is None else v, int):
                pass                                        # an int knob whose None means "follow another key"
            else:
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(old), type(v), k))
        if isinstance(v, dict):
            try:
                _merge_a_into_b(v, b[k])
            except Exception:
                print('Error under config key: {}'.format(k))
                raise
        else:
            b[k] = v


def cfg_from_file(filename):
    """Load a YAML config file and merge it into the defaults."""
    import yaml
    with open(filename, 'r') as f:
        _merge_a_into_b(AttrDict(yaml.safe_load(f) or {}), __C)


def cfg_from_list(cfg_list):
    """`--set K1 V1 K2 V2 ...` (dotted keys; values parsed as Python literals when possible)."""
    assert len(cfg_list) % 2 == 0
    for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
        path = k.split('.')
        d = __C
        for sub in path[:-1]:
            assert sub in d
            d = d[sub]
        leaf = path[-1]
        assert leaf in d
        try:
            value = ast.literal_eval(v)
        except Exception:
            value = v
        old = d[leaf]
        if isinstance(old, tuple) and isinstance(value, list):
            value = tuple(value)
        if isinstance(old, list) and isinstance(value, tuple):
            value = list(value)
        nullable = leaf in _NULLABLE_KEYS and (value is None or old is None) and isinstance(old if value is None else value, int)
        assert nullable or type(value) == type(old), 'type {} does not match original type {}'.format(type(value), type(old))
        d[leaf] = value
