"""Weights of a Keras SavedModel directory without TensorFlow (beside tflite_import.py).

The reference keeps its current network as a SavedModel (`corintho_ai/model/`, loaded by `keras.models.load_model` at
corintho_ai/python/main.pyx:296 and written back at :322; built at wrapper.py:256-271).  Its variables live in a
TensorBundle: `variables/variables.index`, a table of `name -> BundleEntryProto {dtype, shape, shard, offset, size,
crc32c}` in the LevelDB sorted-table format (prefix-compressed entries in blocks, a block index, a 48-byte footer), and
`variables/variables.data-0000N-of-0000M`, the tensors' raw bytes.  Keras names a layer's variables
`layer_with_weights-<i>/<kernel|bias|gamma|beta|moving_mean|moving_variance>/.ATTRIBUTES/VARIABLE_VALUE`, i counting the
layers that have weights in creation order -- for the reference's network Dense 0, BatchNormalization 0, ..., Dense 11,
BatchNormalization 11, the value head Dense(1), the policy head Dense(96).

Unlike the TFLite checkpoints (BatchNorm folded into the next layer by the converter) the SavedModel holds BatchNorm as
Keras does: gamma, beta and the moving statistics -- the explicit-BN layout of nets.py (`get_weights()` order).

Only what these files use is implemented: uncompressed blocks, float32 / int64 tensors, little-endian hosts."""
import os
import struct

import numpy as np

from . import nets

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT64 = 1, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT64: np.dtype("<i8"), 3: np.dtype("<i4"), 2: np.dtype("<f8")}


class SavedModelFormatError(ValueError):
    pass


def _varint(buf, pos):
    out = shift = 0
    while True:
        if pos >= len(buf):
            raise SavedModelFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 63:
            raise SavedModelFormatError("varint too long")


def _block(buf, offset, size):
    """the contents of the block at (offset, size): its entries as (key, value) pairs, keys rebuilt from the prefix
    compression.  A block is followed by one byte of compression type and four of CRC."""
    if offset + size + 5 > len(buf):
        raise SavedModelFormatError("block beyond the end of the file")
    if buf[offset + size] != 0:
        raise SavedModelFormatError("compressed table block (type %d): not supported" % buf[offset + size])
    blk = buf[offset:offset + size]
    if size < 4:
        raise SavedModelFormatError("block too small")
    n_restarts = struct.unpack_from("<I", blk, size - 4)[0]
    end = size - 4 - 4 * n_restarts
    if end < 0:
        raise SavedModelFormatError("bad restart array")
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        if shared > len(key) or pos + non_shared + vlen > end:
            raise SavedModelFormatError("bad table entry")
        key = key[:shared] + bytes(blk[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(blk[pos:pos + vlen])))
        pos += vlen
    return out


def _entry(value):
    """BundleEntryProto: 1 dtype, 2 shape {2: dim {1: size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32)"""
    e = {"dtype": 0, "shape": [], "shard": 0, "offset": 0, "size": 0}
    pos = 0
    while pos < len(value):
        tag, pos = _varint(value, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(value, pos)
            if field == 1:
                e["dtype"] = v
            elif field == 3:
                e["shard"] = v
            elif field == 4:
                e["offset"] = v
            elif field == 5:
                e["size"] = v
        elif wire == 5:
            pos += 4
        elif wire == 1:
            pos += 8
        elif wire == 2:
            n, pos = _varint(value, pos)
            sub = value[pos:pos + n]
            pos += n
            if field == 2:  # TensorShapeProto
                sp = 0
                while sp < len(sub):
                    t, sp = _varint(sub, sp)
                    if t & 7 == 2:
                        m, sp = _varint(sub, sp)
                        dim = sub[sp:sp + m]
                        sp += m
                        if t >> 3 == 2:  # Dim
                            dp, size = 0, 0
                            while dp < len(dim):
                                dt, dp = _varint(dim, dp)
                                if dt & 7 == 0:
                                    dv, dp = _varint(dim, dp)
                                    if dt >> 3 == 1:
                                        size = dv
                                elif dt & 7 == 2:
                                    k, dp = _varint(dim, dp)
                                    dp += k
                                else:
                                    raise SavedModelFormatError("unexpected wire type in a shape")
                            e["shape"].append(size)
                    elif t & 7 == 0:
                        _, sp = _varint(sub, sp)
                    else:
                        raise SavedModelFormatError("unexpected wire type in a shape")
        else:
            raise SavedModelFormatError("unexpected wire type %d in a bundle entry" % wire)
    return e


def read_index(index_bytes):
    """-> {name: entry dict} of a TensorBundle index file (the header entry under the empty key is left out)"""
    buf = memoryview(index_bytes)
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise SavedModelFormatError("not a TensorBundle index (no table footer)")
    foot = buf[len(buf) - 48:len(buf) - 8]
    pos = 0
    _, pos = _varint(foot, pos)  # metaindex handle
    _, pos = _varint(foot, pos)
    ioff, pos = _varint(foot, pos)
    isize, pos = _varint(foot, pos)
    entries = {}
    for _, handle in _block(buf, ioff, isize):
        hp = 0
        boff, hp = _varint(handle, hp)
        bsize, hp = _varint(handle, hp)
        for key, value in _block(buf, boff, bsize):
            if key:
                entries[key.decode("utf-8")] = _entry(value)
    return entries


def read_tensor_bundle(prefix):
    """-> {name: ndarray} of every numeric tensor of the bundle `<prefix>.index` + `<prefix>.data-*`"""
    with open(prefix + ".index", "rb") as f:
        entries = read_index(f.read())
    shards = {}
    out = {}
    n_shards = 1 + max([e["shard"] for e in entries.values()] or [0])
    for name, e in entries.items():
        if e["dtype"] not in _DTYPES:
            continue  # (the object graph is a string tensor)
        if e["shard"] not in shards:
            with open("%s.data-%05d-of-%05d" % (prefix, e["shard"], n_shards), "rb") as f:
                shards[e["shard"]] = f.read()
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        dt = _DTYPES[e["dtype"]]
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
            raise SavedModelFormatError("%s: %d bytes for shape %s" % (name, e["size"], e["shape"]))
        out[name] = np.frombuffer(raw, dt).reshape(e["shape"]).copy()
    return out


_SUFFIX = "/.ATTRIBUTES/VARIABLE_VALUE"


def mlp12x100_from_savedmodel(model_dir):
    """-> flat float32 weights (nets.MLP_NUM_WEIGHTS, explicit BatchNorm: nets.py's layout) of the reference's network
    stored as a Keras SavedModel directory (`corintho_ai/model`).  Raises SavedModelFormatError when the variables are
    not those of the 12 x 100 MLP of wrapper.py:256-271."""
    t = read_tensor_bundle(os.path.join(model_dir, "variables", "variables"))

    def var(layer, what, shape):
        key = "layer_with_weights-%d/%s%s" % (layer, what, _SUFFIX)
        if key not in t:
            raise SavedModelFormatError("no variable %s" % key)
        a = t[key]
        if list(a.shape) != list(shape) or a.dtype != np.float32:
            raise SavedModelFormatError("%s: shape %s, expected %s" % (key, list(a.shape), list(shape)))
        return a

    parts = []
    n_in = nets.GAME_STATE_SIZE
    for layer in range(12):
        parts += [var(2 * layer, "kernel", (n_in, 100)).ravel(), var(2 * layer, "bias", (100,))]
        parts += [var(2 * layer + 1, w, (100,)) for w in ("gamma", "beta", "moving_mean", "moving_variance")]
        n_in = 100
    parts += [var(24, "kernel", (100, 1)).ravel(), var(24, "bias", (1,))]
    parts += [var(25, "kernel", (100, nets.NUM_MOVES)).ravel(), var(25, "bias", (nets.NUM_MOVES,))]
    w = np.concatenate(parts).astype(np.float32)
    if w.size != nets.MLP_NUM_WEIGHTS:
        raise SavedModelFormatError("unexpected number of weights %d" % w.size)
    return w
