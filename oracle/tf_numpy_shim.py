"""TEST INFRASTRUCTURE ONLY.  A numpy-backed stand-in for the handful of `tf.*` calls the reference's USE_E2E_TF
functions make (lib/layer_utils/snippets.py:32-49 generate_anchors_pre_tf, proposal_layer.py:56-84 proposal_layer_tf,
proposal_top_layer.py:58-85 proposal_top_layer_tf, model/bbox_transform.py:82-113), so that the reference's OWN function
bodies run here without TensorFlow and pin the graph wiring of oracle/frcnn_oracle.py's TF-mode restatement: which
values are truncated to int32, what is decoded / clipped / gathered in which order, output shapes and dtypes.

Element-wise ops keep TensorFlow's dtype rules for the dtypes that occur (float32 stays float32, int32 stays int32).
Two TensorFlow KERNELS are not the reference's code and are restated from TensorFlow r1.2's sources in
frcnn_oracle.py (tf_non_max_suppression, tf_top_k): those two are "parity unpinned" (TensorFlow is not installable
offline); everything around them is pinned through this shim.
"""
import types

import numpy as np

int32, float32 = np.int32, np.float32


def _a(x):
    return np.asarray(x)


def build(ora):
    tf = types.ModuleType("tensorflow")
    tf.int32, tf.float32 = int32, float32
    tf.range = lambda n: np.arange(int(n), dtype=int32)
    tf.meshgrid = lambda x, y: tuple(np.meshgrid(x, y))
    tf.reshape = lambda t, shape: np.reshape(_a(t), [int(s) for s in shape])
    tf.transpose = lambda t, perm=None: np.transpose(_a(t), perm)
    tf.stack = lambda ts, axis=0: np.stack([_a(t) for t in ts], axis=axis)
    tf.multiply = lambda a, b: _a(a) * _a(b)
    tf.add = lambda a, b: _a(a) + _a(b)
    tf.subtract = lambda a, b: _a(a) - _a(b)
    tf.exp = lambda a: np.exp(_a(a))
    tf.maximum = lambda a, b: np.maximum(_a(a), np.asarray(b, dtype=_a(a).dtype))
    tf.minimum = lambda a, b: np.minimum(_a(a), np.asarray(b, dtype=_a(a).dtype))
    tf.cast = lambda t, dtype: _a(t).astype(dtype)            # float -> int32 truncates toward zero, like tf.cast
    tf.constant = lambda v, dtype=None: _a(v).astype(dtype) if dtype is not None else _a(v)
    tf.to_float = lambda t: _a(t).astype(float32)
    tf.gather = lambda t, idx: _a(t)[_a(idx)]
    tf.zeros = lambda shape, dtype=float32: np.zeros([int(s) for s in shape], dtype=dtype)
    tf.shape = lambda t: np.array(_a(t).shape, dtype=int32)
    tf.concat = lambda ts, axis: np.concatenate([_a(t) for t in ts], axis=axis)
    tf.image = types.SimpleNamespace(
        non_max_suppression=lambda boxes, scores, max_output_size, iou_threshold=0.5:
        ora.tf_non_max_suppression(_a(boxes), _a(scores), int(max_output_size), iou_threshold))
    tf.nn = types.SimpleNamespace(top_k=lambda v, k: ora.tf_top_k(_a(v), int(k)))
    return tf
