"""CPU: TensorFlow V2 checkpoint (tensor bundle) reader / writer without TensorFlow (frcnn_hip/tensor_bundle.py, SURVEY.md 8f
row 1).  TensorFlow is third party and absent: PARITY UNPINNED -- the format facts are asserted on hand-assembled bytes
(independent of the writer) and on published CRC-32C / LevelDB constants; reader and writer are then checked against each other."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd"))
from frcnn_hip import tensor_bundle as tb  # noqa: E402


def test_crc32c_known_answers_and_masking():
    assert tb.crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    assert tb.crc32c(bytes(32)) == 0x8A9136AA and tb.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E and tb.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    big = np.random.RandomState(0).bytes(70001)                       # native slicing-by-8 path == bytewise table path
    c = 0xFFFFFFFF
    t = tb._crc_table()
    for b in big:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    assert tb.crc32c(big) == c ^ 0xFFFFFFFF
    assert tb.crc32c(big[1000:], tb.crc32c(big[:1000])) == tb.crc32c(big)                          # extendable
    # leveldb/tensorflow crc32c::Mask: rotate right by 15, add 0xa282ead8
    assert tb.mask_crc(0) == 0xA282EAD8 and tb.unmask_crc(tb.mask_crc(0xDEADBEEF)) == 0xDEADBEEF
    assert tb.mask_crc(tb.crc32c(b"foo")) != tb.crc32c(b"foo")


def _varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def test_reader_on_hand_assembled_bundle(tmp_path):
    """Bytes laid out here from the format description alone (no call into the writer): one data block with prefix-compressed
    keys and two restart points, an empty metaindex block, an index block, the 48-byte footer."""
    w = np.arange(24, dtype="<f4").reshape(2, 3, 4)
    step = np.array(7, dtype="<i8")
    data = w.tobytes() + step.tobytes()
    (tmp_path / "m.ckpt.data-00000-of-00001").write_bytes(data)

    def entry(dtype, shape, offset, size, raw):
        dims = b"".join(b"\x12" + _varint(len(b"\x08" + _varint(d))) + b"\x08" + _varint(d) for d in shape)
        e = b"\x08" + _varint(dtype) + b"\x12" + _varint(len(dims)) + dims
        if offset:
            e += b"\x20" + _varint(offset)
        e += b"\x28" + _varint(size) + b"\x35" + struct.pack("<I", tb.mask_crc(tb.crc32c(raw)))
        return e

    header = b"\x08\x01" + b"\x1a\x02\x08\x01"                      # num_shards = 1, version { producer = 1 }
    kv = [(b"", header), (b"scope/step", entry(9, (), 96, 8, step.tobytes())), (b"scope/weights", entry(1, (2, 3, 4), 0, 96, w.tobytes()))]
    block, restarts, last = b"", [], b""
    for i, (k, v) in enumerate(kv):
        shared = 0
        if i == 2:                                                    # third key shares the prefix "scope/" with the second
            shared = 6
        else:
            restarts.append(len(block))
        block += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
        last = k
    block += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))

    def with_trailer(b):
        return b + b"\x00" + struct.pack("<I", tb.mask_crc(tb.crc32c(b + b"\x00")))

    f = with_trailer(block)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)
    meta_off = len(f)
    f += with_trailer(meta)
    handle = _varint(0) + _varint(len(block))
    index = _varint(0) + _varint(len(last)) + _varint(len(handle)) + last + handle + struct.pack("<I", 0) + struct.pack("<I", 1)
    index_off = len(f)
    f += with_trailer(index)
    footer = _varint(meta_off) + _varint(len(meta)) + _varint(index_off) + _varint(len(index))
    f += footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    (tmp_path / "m.ckpt.index").write_bytes(f)

    r = tb.BundleReader(str(tmp_path / "m.ckpt"))
    assert r.get_variable_to_shape_map() == {"scope/step": [], "scope/weights": [2, 3, 4]}
    assert r.get_variable_to_dtype_map()["scope/step"] is np.int64 and r.has_tensor("scope/weights") and not r.has_tensor("nope")
    assert np.array_equal(r.get_tensor("scope/weights"), w) and r.get_tensor("scope/step") == 7 and r.get_tensor("scope/step").shape == ()
    with pytest.raises(KeyError):
        r.get_tensor("nope")
    # corruption is detected: one flipped data byte, one flipped index byte
    bad = bytearray(data)
    bad[5] ^= 1
    (tmp_path / "m.ckpt.data-00000-of-00001").write_bytes(bytes(bad))
    with pytest.raises(IOError, match="checksum"):
        tb.BundleReader(str(tmp_path / "m.ckpt")).get_tensor("scope/weights")
    assert np.array_equal(tb.BundleReader(str(tmp_path / "m.ckpt"), verify=False).get_tensor("scope/weights").ravel()[2:], w.ravel()[2:])
    fb = bytearray(f)
    fb[3] ^= 0x40
    (tmp_path / "m.ckpt.index").write_bytes(bytes(fb))
    with pytest.raises(IOError, match="checksum"):
        tb.BundleReader(str(tmp_path / "m.ckpt"))
    (tmp_path / "m.ckpt.index").write_bytes(f[:-8] + b"\x00" * 8)
    with pytest.raises(IOError, match="magic"):
        tb.BundleReader(str(tmp_path / "m.ckpt"))
    snappy = bytearray(f)
    snappy[len(block)] = 1                                            # compression type byte of the data block
    snappy[len(block) + 1:len(block) + 5] = struct.pack("<I", tb.mask_crc(tb.crc32c(block + b"\x01")))
    (tmp_path / "m.ckpt.index").write_bytes(bytes(snappy))
    with pytest.raises(IOError, match="corrupted compressed block contents"):      # the message train_val.py:111 looks for
        tb.BundleReader(str(tmp_path / "m.ckpt"))


def test_writer_reader_round_trip_many_blocks(tmp_path):
    rng = np.random.RandomState(1)
    tensors = {"resnet_v1_101/block3/unit_%d/bottleneck_v1/conv2/weights" % i: rng.randn(3, 3, 4, 8).astype(np.float32) for i in range(1, 24)}
    tensors.update({"resnet_v1_101/conv1/BatchNorm/moving_variance": rng.rand(64).astype(np.float32), "global_step": np.array(70000, dtype=np.int64),
                    "flags": np.array([True, False]), "half": rng.randn(5).astype(np.float16), "empty": np.zeros((0, 4), np.float32),
                    "resnet_v1_101/conv1/weights/Momentum": rng.randn(7, 7, 3, 64).astype(np.float32)})
    prefix = tb.write_bundle(str(tmp_path / "res101_faster_rcnn_iter_70000.ckpt"), tensors)
    assert sorted(os.listdir(str(tmp_path))) == ["res101_faster_rcnn_iter_70000.ckpt.data-00000-of-00001", "res101_faster_rcnn_iter_70000.ckpt.index"]
    pairs = tb.read_table(prefix + ".index")
    assert [k for k, _ in pairs] == sorted(k for k, _ in pairs) and pairs[0][0] == b"" and len(pairs) == len(tensors) + 1
    r = tb.BundleReader(prefix + ".index")                            # the .index path is accepted like the prefix
    assert set(r.get_variable_to_shape_map()) == set(tensors)
    for k, v in tensors.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    with pytest.raises(ValueError):
        tb.write_table(str(tmp_path / "t"), [(b"b", b"1"), (b"a", b"2")])
    with pytest.raises(IOError, match="not found"):
        tb.BundleReader(str(tmp_path / "missing.ckpt"))


def test_session_restore_and_per_network_fixes(tmp_path):
    """train_val.py:177-202 / resnet_v1.py:154-178 / vgg16.py:62-100 / mobilenet_v1.py:253-278 on a pretrained-style bundle."""
    sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd", "lib"))
    from frcnn_hip.runtime import VariableStore
    from nets.mobilenet_v1 import mobilenetv1
    from nets.resnet_v1 import resnetv1
    from nets.vgg16 import vgg16
    rng = np.random.RandomState(2)
    for ctor, stem, extra_fix in ((lambda: resnetv1(num_layers=50), "/conv1", ()), (vgg16, "/conv1/conv1_1", ("/fc6", "/fc7")),
                                  (mobilenetv1, "/Conv2d_0", ())):
        net = ctor()
        net.create_architecture("TEST", 21, tag="ck", anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2))
        sess = VariableStore(seed=3)                                     # the host half of a Session (no GPU needed)
        scope = net._scope
        specs = net.variable_specs()
        fc6_conv, fc7_conv = (7, 7, 512, 4096), (1, 1, 4096, 4096)
        if extra_fix:       # VGG's 120 M fc parameters make this a disk benchmark: shrink fc6/fc7 (the fix logic is shape-generic)
            from frcnn_hip.runtime import VarSpec
            specs[scope + "/fc6/weights"], specs[scope + "/fc6/biases"] = VarSpec((7 * 7 * 512, 8), "he"), VarSpec((8,), "zeros")
            specs[scope + "/fc7/weights"], specs[scope + "/fc7/biases"] = VarSpec((8, 8), "he"), VarSpec((8,), "zeros")
            fc6_conv, fc7_conv = (7, 7, 512, 8), (1, 1, 8, 8)
        sess.init_variables(specs)
        # an "ImageNet" checkpoint: backbone variables only (no RPN / heads), RGB stem, VGG fc layers stored as conv filters
        ck = {k: (v * np.float32(2) + np.float32(1)) for k, v in sess.variables.items()       # != the initial values, cheap for VGG's 138 M
              if "rpn" not in k and "cls_score" not in k and "bbox_pred" not in k}
        if extra_fix:
            ck[scope + "/fc6/weights"] = ck[scope + "/fc6/weights"].reshape(fc6_conv)
            ck[scope + "/fc7/weights"] = ck[scope + "/fc7/weights"].reshape(fc7_conv)
        ck["global_step"] = np.array(0, dtype=np.int64)
        prefix = tb.write_bundle(str(tmp_path / (scope + ".ckpt")), ck)
        before = dict(sess.variables)                                                 # restore() rebinds entries, never mutates them
        shapes = tb.BundleReader(prefix).get_variable_to_shape_map()
        names = net.get_variables_to_restore(list(sess.variables), shapes)
        fix_names = [scope + stem + "/weights"] + [scope + t + "/weights" for t in extra_fix]
        assert not set(fix_names) & set(names) and all(n in ck for n in names) and "global_step" not in names
        sess.restore(prefix, names)
        assert sorted(net.fix_variables(sess, prefix)) == sorted(fix_names)
        for k in sess.variables:
            if k in fix_names:
                continue
            want = ck[k] if k in ck else before[k]                               # heads keep their initialisation
            assert np.array_equal(sess.variables[k], want), k
        stem_w = ck[scope + stem + "/weights"]
        if ctor is mobilenetv1:
            stem_w = stem_w / np.float32(127.5)
        assert np.array_equal(sess.variables[scope + stem + "/weights"], stem_w[:, :, ::-1, :])       # RGB -> BGR
        for t in extra_fix:
            assert np.array_equal(sess.variables[scope + t + "/weights"], ck[scope + t + "/weights"].reshape(before[scope + t + "/weights"].shape))
        # plain restore of a full (trained) checkpoint: tools/test_net.py:110-114
        full = str(tmp_path / (scope + "_full.ckpt"))
        sess.save(full, {"global_step": np.array(5, dtype=np.int64)})
        sess2 = VariableStore(seed=9)
        sess2.init_variables(specs)
        sess2.restore(full)
        assert all(np.array_equal(sess2.variables[k], sess.variables[k]) for k in sess.variables)
        with pytest.raises(KeyError):
            sess2.restore(prefix)                                                # the ImageNet checkpoint has no RPN / head variables


# ---------------------------------------------------------------------------------------------------- V1 checkpoints + snappy
def _pb(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _pbv(field, v):
    return _varint(field << 3) + _varint(v)


def _shape_pb(shape):
    return b"".join(_pb(2, _pbv(1, d)) for d in shape)


def _snappy_literals(data, chunk=60):
    """A valid Snappy stream made of literals only (what a compressor emits for incompressible data)."""
    out = _varint(len(data))
    for i in range(0, len(data), chunk):
        part = data[i:i + chunk]
        out += bytes([(len(part) - 1) << 2]) + part
    return out


def _write_v1(path, tensors, slices=None, snappy=False):
    """Test-side writer of the V1 layout (SavedTensorSlices protos in an SSTable), assembled from the format description."""
    metas, entries = b"", []
    for name, a in sorted(tensors.items()):
        dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 9, np.dtype(np.int32): 3}[a.dtype]
        parts = (slices or {}).get(name, [tuple((0, None) for _ in a.shape)])
        metas += _pb(1, _pb(1, name.encode()) + _pb(2, _shape_pb(a.shape)) + _pbv(3, dt))
        for k, ext in enumerate(parts):
            idx = tuple(slice(st, None if ln is None else st + ln) for st, ln in ext)
            sub = np.ascontiguousarray(a[idx]) if ext else a
            ext_pb = b"".join(_pb(1, (_pbv(1, st) if st else b"") + (_pbv(2, ln) if ln is not None else b"")) for st, ln in ext)
            if a.dtype == np.float32:
                vals = _pb(5, sub.astype("<f4").tobytes())                       # packed float_val, as SaveData/Fill writes it
            elif a.dtype == np.int64:
                vals = _pb(10, b"".join(_varint(int(v) & ((1 << 64) - 1)) for v in sub.ravel()))
            else:
                vals = _pb(4, sub.astype("<i4").tobytes())                       # tensor_content
            tensor = _pbv(1, dt) + _pb(2, _shape_pb(sub.shape)) + vals
            entries.append((b"\\x00" + name.encode() + b"\\x00" + bytes([k + 1]), _pb(2, _pb(1, name.encode()) + _pb(2, ext_pb) + _pb(3, tensor))))
    items = [(b"", _pb(1, metas))] + sorted(entries)
    if not snappy:
        tb.write_table(path, items)
        return
    # one snappy-compressed data block (type byte 1), hand-assembled like test_reader_on_hand_assembled_bundle
    block, restarts = b"", []
    for k, v in items:
        restarts.append(len(block))
        block += _varint(0) + _varint(len(k)) + _varint(len(v)) + k + v
    block += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    comp = _snappy_literals(block)

    def trailer(b, t):
        return b + bytes([t]) + struct.pack("<I", tb.mask_crc(tb.crc32c(b + bytes([t]))))

    f = trailer(comp, 1)
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)
    meta_off = len(f)
    f += trailer(meta, 0)
    handle = _varint(0) + _varint(len(comp))
    last = items[-1][0]
    index = _varint(0) + _varint(len(last)) + _varint(len(handle)) + last + handle + struct.pack("<I", 0) + struct.pack("<I", 1)
    index_off = len(f)
    f += trailer(index, 0)
    footer = _varint(meta_off) + _varint(len(meta)) + _varint(index_off) + _varint(len(index))
    f += footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    with open(path, "wb") as fh:
        fh.write(f)


def test_snappy_uncompress_known_streams():
    # google/snappy format_description.txt: literals, copies with 1/2/4-byte offsets, overlapping (run-length) copies
    assert tb.snappy_uncompress(bytes([11]) + bytes([10 << 2]) + b"hello world") == b"hello world"
    assert tb.snappy_uncompress(bytes([9, 2 << 2]) + b"abc" + bytes([((6 - 4) << 2) | 1, 3])) == b"abcabcabc"        # copy-1: len 6, offset 3
    assert tb.snappy_uncompress(bytes([10, 0]) + b"x" + bytes([((9 - 1) << 2) | 2, 1, 0])) == b"x" * 10               # copy-2, overlapping run
    assert tb.snappy_uncompress(bytes([8, 3 << 2]) + b"wxyz" + bytes([((4 - 1) << 2) | 3, 4, 0, 0, 0])) == b"wxyzwxyz"  # copy-4
    long_lit = bytes(range(256)) * 2
    assert tb.snappy_uncompress(_varint(512) + bytes([61 << 2]) + struct.pack("<H", 511) + long_lit) == long_lit      # 2-byte literal length
    assert tb.snappy_uncompress(_snappy_literals(long_lit)) == long_lit
    for bad in (bytes([5, 0]) + b"x", bytes([3, ((4 - 1) << 2) | 2, 9, 0]), bytes([200])):                             # short / offset before start / cut header
        with pytest.raises(IOError, match="corrupted compressed block contents"):
            tb.snappy_uncompress(bad)


@pytest.mark.parametrize("snappy", [False, True], ids=["plain", "snappy"])
def test_v1_checkpoint_reader(tmp_path, snappy):
    rng = np.random.RandomState(4)
    tensors = {"resnet_v1_101/conv1/weights": rng.randn(7, 7, 3, 8).astype(np.float32),
               "resnet_v1_101/block1/unit_1/bottleneck_v1/conv1/BatchNorm/gamma": rng.rand(8).astype(np.float32),
               "partitioned/embedding": rng.randn(10, 6).astype(np.float32),
               "global_step": np.array(123456789012, dtype=np.int64), "ids": np.arange(-3, 4, dtype=np.int32)}
    slices = {"partitioned/embedding": [((0, 4), (0, None)), ((4, 6), (0, None))]}          # two row partitions
    path = str(tmp_path / "resnet_v1_101.ckpt")
    _write_v1(path, tensors, slices, snappy=snappy)
    r = tb.open_checkpoint(path)
    assert isinstance(r, tb.CheckpointReaderV1)
    assert r.get_variable_to_shape_map() == {k: list(v.shape) for k, v in tensors.items()}
    for k, v in tensors.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    assert not r.has_tensor("nope")
    with pytest.raises(KeyError):
        r.get_tensor("nope")
    # the factory picks V2 when an index file exists, and Session.restore reads either
    tb.write_bundle(str(tmp_path / "v2.ckpt"), tensors)
    assert isinstance(tb.open_checkpoint(str(tmp_path / "v2.ckpt")), tb.BundleReader)
    from frcnn_hip.runtime import VariableStore
    st = VariableStore()
    st.variables["resnet_v1_101/conv1/weights"] = np.zeros((7, 7, 3, 8), np.float32)
    st.restore(path, ["resnet_v1_101/conv1/weights"])
    assert np.array_equal(st.variables["resnet_v1_101/conv1/weights"], tensors["resnet_v1_101/conv1/weights"])
    with pytest.raises(IOError, match="not found"):
        tb.open_checkpoint(str(tmp_path / "missing.ckpt"))


def test_find_previous_and_remove_snapshot(tmp_path):
    """train_val.py:155-175, 235-256: newest snapshot pair is found (STEPSIZE+1 extras skipped), old ones are pruned."""
    import time
    sys.path.insert(0, os.path.join(ROOT, "tf-faster-rcnn_amd", "lib"))
    from model.config import cfg
    from model.train_val import find_previous, remove_snapshot
    old = (cfg.TRAIN.STEPSIZE, cfg.TRAIN.SNAPSHOT_KEPT)
    cfg.TRAIN.STEPSIZE, cfg.TRAIN.SNAPSHOT_KEPT = [30], 2
    try:
        assert find_previous(str(tmp_path)) == (0, [], [])
        now = time.time()
        for k, it in enumerate((10, 20, 31, 40)):                                 # 31 = STEPSIZE + 1: not a resume point
            base = str(tmp_path / ("res101_faster_rcnn_iter_%d" % it))
            for suffix in (".ckpt.index", ".ckpt.data-00000-of-00001", ".pkl"):
                with open(base + suffix, "wb") as f:
                    f.write(b"x")
                os.utime(base + suffix, (now + k, now + k))
        n, nfiles, sfiles = find_previous(str(tmp_path))
        assert n == 3 and [os.path.basename(s) for s in sfiles] == ["res101_faster_rcnn_iter_%d.ckpt" % i for i in (10, 20, 40)]
        assert [os.path.basename(s) for s in nfiles] == ["res101_faster_rcnn_iter_%d.pkl" % i for i in (10, 20, 40)]
        remove_snapshot(nfiles, sfiles)
        assert len(nfiles) == 2 and len(sfiles) == 2
        left = sorted(os.listdir(str(tmp_path)))
        assert not any("iter_10" in f for f in left) and any("iter_20.ckpt.index" in f for f in left) and any("iter_31" in f for f in left)
    finally:
        cfg.TRAIN.STEPSIZE, cfg.TRAIN.SNAPSHOT_KEPT = old
