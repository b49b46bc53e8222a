"""TensorFlow checkpoint reader / writer without TensorFlow: V2 "tensor bundle" read + write, V1
single-file checkpoints read (CheckpointReaderV1, at the end of this file); open_checkpoint() picks the format.

What the reference gets from `pywrap_tensorflow.NewCheckpointReader` / `tf.train.Saver` (lib/model/train_val.py:105-114,
177-202, tools/test_net.py:110-114): `<prefix>.index` + `<prefix>.data-SSSSS-of-NNNNN`.

Format (third-party: TensorFlow r1.2, tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, core/lib/io/{table,block,
format}.cc, core/protobuf/tensor_bundle.proto; restated -- PARITY UNPINNED: neither TensorFlow nor a checkpoint file exists in
this container or in the reference tree, so reader and writer are checked against each other and against hand-assembled
bytes only):

  .index  = an SSTable (LevelDB table format):
      [data block]* [metaindex block] [index block] [footer, 48 bytes]
      block    = entries | restart offsets (u32 LE each) | num_restarts (u32 LE);   then a 5-byte trailer on disk:
                 compression type (0 none, 1 snappy) + masked crc32c(block + type) (u32 LE)
      entry    = varint32 shared | varint32 non_shared | varint32 value_len | key[shared:] | value   (prefix compression)
      index    = key >= last key of a data block  ->  BlockHandle(varint64 offset, varint64 size)
      footer   = metaindex BlockHandle, index BlockHandle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (u64 LE)
    key ""   -> BundleHeaderProto { num_shards = 1; endianness = 2; version = 3 { producer = 1 } }
    key name -> BundleEntryProto  { dtype = 1; shape = 2 { dim { size = 1 } }; shard_id = 3; offset = 4; size = 5;
                                    crc32c = 6 (fixed32, masked); slices = 7 }
  .data-* = the raw little-endian tensor bytes, entry by entry in key order.

The bundle writer TensorFlow uses does not compress blocks; Snappy blocks (type 1: old V1 files) are decoded by the
native helper frcnn_snappy_uncompress.
"""
import ctypes
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16,
          19: np.float16}
_DT_OF = {np.dtype(v): k for k, v in DTYPES.items()}


# ------------------------------------------------------------------------------------------------ crc32c (Castagnoli)
def _crc_table():
    t = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like object.  Large buffers go through the native helper in libfrcnn_hip.so (host code, slicing
    by 8); small ones (index blocks) through a Python table so the reader also works before the library is built."""
    mv = memoryview(data).cast("B")
    if mv.nbytes >= 4096:
        try:
            from . import lib
            buf = np.frombuffer(mv, dtype=np.uint8)
            return int(lib().frcnn_crc32c(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.size), ctypes.c_uint32(crc)))
        except (ImportError, OSError):
            pass
    global _TABLE
    if _TABLE is None:
        _TABLE = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in mv.tobytes():
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def snappy_uncompress(data):
    """Snappy raw format (compressed table blocks of V1 checkpoints) through the native helper frcnn_snappy_uncompress."""
    mv = bytes(data)
    n = shift = pos = 0
    while True:                                  # varint32 length header
        if pos >= len(mv):
            raise IOError("corrupted compressed block contents")
        b = mv[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            break
        shift += 7
    from . import lib
    out = ctypes.create_string_buffer(max(n, 1))
    got = lib().frcnn_snappy_uncompress(mv, ctypes.c_size_t(len(mv)), out, ctypes.c_size_t(n))
    if got != n:
        raise IOError("corrupted compressed block contents")           # the message train_val.py:111 looks for
    return out.raw[:n]


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - _MASK_DELTA) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf wire
def _put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _pb_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message; value = int (varint / fixed) or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _pb_varint(field, v):
    return _put_varint(field << 3) + _put_varint(v)


def _pb_bytes(field, b):
    return _put_varint((field << 3) | 2) + _put_varint(len(b)) + b


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _encode_entry(dtype, shape, shard_id, offset, size, crc_masked):
    dims = b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)
    out = _pb_varint(1, dtype) + _pb_bytes(2, dims)
    if shard_id:
        out += _pb_varint(3, shard_id)
    if offset:
        out += _pb_varint(4, offset)
    if size:
        out += _pb_varint(5, size)
    return out + _put_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)


def _decode_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=0, slices=0)
    for field, _, v in _pb_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            for f2, _, dim in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, sv in _pb_fields(dim):
                        if f3 == 1:
                            size = _signed64(sv)
                    e["shape"].append(size)
                elif f2 == 3 and dim:
                    raise ValueError("tensor of unknown rank in checkpoint")
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v
        elif field == 7:
            e["slices"] += 1
    return e


# ------------------------------------------------------------------------------------------------ SSTable
def _read_block(data, offset, size, verify):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
            raise IOError("checkpoint index: block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        return snappy_uncompress(raw)
    if ctype != 0:
        raise IOError("checkpoint index: unknown block compression type %d" % ctype)
    return raw


def _block_entries(block):
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of an SSTable file, in key order."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise IOError("%s is not an SSTable (bad magic number): not a V2 checkpoint index" % path)
    footer = data[len(data) - 48:]
    _, pos = _get_varint(footer, 0)                # metaindex handle (unused: the bundle writes no filter)
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, p = _get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


class _BlockBuilder(object):
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.counter, self.last, self.interval = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.counter < self.interval:
            m = min(len(self.last), len(key))
            while shared < m and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.counter += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=4096):
    """items: iterable of (key bytes, value bytes) in strictly increasing key order."""
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)

    def emit(block_bytes):
        off = len(out)
        out.extend(block_bytes)
        out.append(0)                                                     # kNoCompression
        out.extend(struct.pack("<I", mask_crc(crc32c(block_bytes + b"\x00"))))
        return _put_varint(off) + _put_varint(len(block_bytes))

    cur, last_key, prev = _BlockBuilder(), None, None
    for key, value in items:
        if prev is not None and not key > prev:
            raise ValueError("table keys must be strictly increasing")
        prev = key
        cur.add(key, value)
        last_key = key
        if cur.size() >= block_size:
            index.add(last_key, emit(cur.finish()))
            cur = _BlockBuilder()
    if cur.counter or last_key is None:
        index.add(last_key if last_key is not None else b"", emit(cur.finish()))
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------------------------------------ bundle
def _shard_name(prefix, shard, num):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num)


class BundleReader(object):
    """`pywrap_tensorflow.NewCheckpointReader(prefix)` look-alike: has_tensor / get_tensor / get_variable_to_shape_map /
    get_variable_to_dtype_map."""

    def __init__(self, prefix, verify=True):
        if prefix.endswith(".index"):
            prefix = prefix[:-len(".index")]
        self.prefix, self.verify = prefix, verify
        if not os.path.isfile(prefix + ".index"):
            hint = " (a V1 single-file checkpoint is not a tensor bundle)" if os.path.isfile(prefix) else ""
            raise IOError("checkpoint index %s.index not found%s" % (prefix, hint))
        self.entries, self.num_shards = {}, 1
        for key, value in read_table(prefix + ".index", verify):
            if key == b"":
                for field, _, v in _pb_fields(value):
                    if field == 1:
                        self.num_shards = v
                    elif field == 2 and v != 0:
                        raise IOError("big-endian checkpoint")
                continue
            e = _decode_entry(value)
            if e["dtype"] not in DTYPES:
                e["unsupported"] = True
            self.entries[key.decode("utf-8")] = e
        self._shards = {}

    def has_tensor(self, name):
        return name in self.entries

    def get_variable_to_shape_map(self):
        return {k: list(e["shape"]) for k, e in self.entries.items()}

    def get_variable_to_dtype_map(self):
        return {k: DTYPES.get(e["dtype"]) for k, e in self.entries.items()}

    def _shard(self, i):
        if i not in self._shards:
            self._shards[i] = np.memmap(_shard_name(self.prefix, i, self.num_shards), dtype=np.uint8, mode="r")
        return self._shards[i]

    def get_tensor(self, name):
        if name not in self.entries:
            raise KeyError("Key %s not found in checkpoint" % name)
        e = self.entries[name]
        if e.get("unsupported"):
            raise NotImplementedError("tensor %s has dtype enum %d (strings / resources are not numeric weights)" % (name, e["dtype"]))
        if e["slices"]:
            raise NotImplementedError("tensor %s is stored as %d slices (partitioned variable)" % (name, e["slices"]))
        raw = self._shard(e["shard_id"])[e["offset"]:e["offset"] + e["size"]]
        dt = np.dtype(DTYPES[e["dtype"]])
        if int(np.prod(e["shape"], dtype=np.int64)) * dt.itemsize != e["size"]:
            raise IOError("tensor %s: %d bytes stored for shape %s" % (name, e["size"], e["shape"]))
        if self.verify and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise IOError("tensor %s: data checksum mismatch" % name)
        return np.frombuffer(raw.tobytes(), dtype=dt.newbyteorder("<")).reshape(e["shape"]).astype(dt, copy=True)     # writable, native order


def write_bundle(prefix, tensors):
    """tensors: {name: ndarray}.  One data shard, entries in key order (what `tf.train.Saver.save` produces for an
    un-sharded saver: train_val.py:58-100 snapshot)."""
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    items = [(b"", _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1)))]        # num_shards 1, LITTLE endian (default), version.producer 1
    offset = 0
    with open(_shard_name(prefix, 0, 1), "wb") as f:
        for name in names:
            a = np.asarray(tensors[name])
            a = a if a.flags.c_contiguous else a.copy()                   # (ascontiguousarray would turn a scalar into shape [1])
            if a.dtype not in _DT_OF:
                raise TypeError("%s: dtype %s cannot be stored" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            items.append((name.encode("utf-8"), _encode_entry(_DT_OF[a.dtype], a.shape, 0, offset, len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + ".index", items)
    return prefix



# ------------------------------------------------------------------------------------------------ V1 checkpoints
class CheckpointReaderV1(object):
    """TensorFlow V1 checkpoint (one file, what `tf.train.Saver` wrote before 1.0 -- the slim ImageNet `resnet_v1_101.ckpt`,
    `vgg_16.ckpt`, `mobilenet_v1_1.0_224.ckpt` the reference starts training from, train_val.py:177-190).

    Format (tensorflow/core/util/{tensor_slice_writer,saved_tensor_slice_util}.cc, saved_tensor_slice.proto; restated,
    PARITY UNPINNED like the V2 reader): the same SSTable container, blocks optionally Snappy-compressed; key "" ->
    SavedTensorSlices{meta = 1: SavedTensorSliceMeta{tensor = 1: {name = 1, shape = 2, type = 3, slice = 4}}}; every other
    key -> SavedTensorSlices{data = 2: SavedSlice{name = 1, slice = 2: TensorSliceProto{extent = 1: {start = 1, length = 2}},
    data = 3: TensorProto{dtype = 1, tensor_shape = 2, tensor_content = 4, float_val = 5 (packed), double_val = 6,
    int_val = 7, int64_val = 10, bool_val = 11, half_val = 13}}}.  Slices of a partitioned variable are assembled."""

    _VAL_FIELDS = {5: ("<f4", False), 6: ("<f8", False), 7: (None, True), 10: (None, True), 11: (None, True), 13: (None, True)}

    def __init__(self, path, verify=True):
        self.prefix = path
        self._shapes, self._dtypes, self._slices = {}, {}, {}
        for key, value in read_table(path, verify):
            for field, _, v in _pb_fields(value):
                if field == 1 and key == b"":
                    self._read_meta(v)
                elif field == 2:
                    self._index_slice(v)

    def _read_meta(self, meta):
        for field, _, t in _pb_fields(meta):
            if field != 1:
                continue
            name, shape, dtype = None, [], 0
            for f2, _, v in _pb_fields(t):
                if f2 == 1:
                    name = v.decode("utf-8")
                elif f2 == 2:
                    shape = _decode_shape(v)
                elif f2 == 3:
                    dtype = v
            self._shapes[name], self._dtypes[name] = shape, dtype

    def _index_slice(self, saved_slice):
        name, extents, tensor = None, [], None
        for field, _, v in _pb_fields(saved_slice):
            if field == 1:
                name = v.decode("utf-8")
            elif field == 2:
                for f2, _, ext in _pb_fields(v):
                    if f2 == 1:
                        start, length = 0, None
                        for f3, _, x in _pb_fields(ext):
                            if f3 == 1:
                                start = _signed64(x)
                            elif f3 == 2:
                                length = _signed64(x)
                        extents.append((start, length))
            elif field == 3:
                tensor = v
        self._slices.setdefault(name, []).append((extents, tensor))

    def has_tensor(self, name):
        return name in self._shapes

    def get_variable_to_shape_map(self):
        return {k: list(v) for k, v in self._shapes.items()}

    def get_variable_to_dtype_map(self):
        return {k: DTYPES.get(v) for k, v in self._dtypes.items()}

    @staticmethod
    def _tensor_values(tensor, dt):
        """Flat values of a TensorProto in dtype dt (tensor_content, or the packed / repeated *_val field)."""
        shape, chunks, ints = [], [], []
        for field, wt, v in _pb_fields(tensor):
            if field == 2:
                shape = _decode_shape(v)
            elif field == 4:
                chunks.append(np.frombuffer(v, dtype=dt.newbyteorder("<")))
            elif field in (5, 6) and wt == 2:
                chunks.append(np.frombuffer(v, dtype="<f4" if field == 5 else "<f8"))
            elif field in (5, 6):                                              # unpacked scalar: fixed32 / fixed64 bits
                chunks.append(np.array([v], dtype="<u4" if field == 5 else "<u8").view("<f4" if field == 5 else "<f8"))
            elif field in (7, 10, 11, 13):
                if wt == 2:
                    pos = 0
                    while pos < len(v):
                        x, pos = _get_varint(v, pos)
                        ints.append(_signed64(x))
                else:
                    ints.append(_signed64(v))
        if ints:
            arr = np.array(ints, dtype=np.int64)
            chunks.append(arr.astype(np.uint16).view(np.float16) if dt == np.float16 else arr)
        flat = np.concatenate(chunks) if chunks else np.zeros((0,), dtype=dt)
        return shape, flat.astype(dt, copy=False)

    def get_tensor(self, name):
        if name not in self._shapes:
            raise KeyError("Key %s not found in checkpoint" % name)
        if self._dtypes[name] not in DTYPES:
            raise NotImplementedError("tensor %s has dtype enum %d" % (name, self._dtypes[name]))
        dt = np.dtype(DTYPES[self._dtypes[name]])
        full = tuple(self._shapes[name])
        out = np.zeros(full, dtype=dt)
        for extents, tensor in self._slices.get(name, []):
            shape, flat = self._tensor_values(tensor, dt)
            index = tuple(slice(st, None if ln is None else st + ln) for st, ln in extents) if extents else ()
            region = out[index] if index else out
            n = int(np.prod(region.shape, dtype=np.int64))
            if flat.size == 1 and n > 1:
                flat = np.full((n,), flat[0], dtype=dt)                      # TensorProto's "repeat the last value" compression
            if flat.size != n:
                raise IOError("tensor %s: slice holds %d values for a region of %s" % (name, flat.size, list(region.shape)))
            region[...] = flat.reshape(region.shape)
        return out


def _decode_shape(buf):
    dims = []
    for f, _, dim in _pb_fields(buf):
        if f == 2:
            size = 0
            for f3, _, sv in _pb_fields(dim):
                if f3 == 1:
                    size = _signed64(sv)
            dims.append(size)
    return dims


def open_checkpoint(path, verify=True):
    """`pywrap_tensorflow.NewCheckpointReader(path)`: a V2 bundle (`<path>.index` + data shards) or a V1 single file."""
    if path.endswith(".index") or os.path.isfile(path + ".index"):
        return BundleReader(path, verify=verify)
    if os.path.isfile(path):
        return CheckpointReaderV1(path, verify=verify)
    raise IOError("checkpoint %s not found (neither %s.index nor a V1 file)" % (path, path))
