"""Test-side writer of TFLite flatbuffers shaped like the reference's checkpoints (a chain of
FULLY_CONNECTED operators with fused ReLU, BatchNormalization folded into the next layer, a
TANH value head and a SOFTMAX policy head), so that corintho_ai_amd/tflite_import.py can be
tested without any reference file.  Only the schema fields the reader uses are written
(tensorflow/lite/schema/schema.fbs)."""
import struct

import numpy as np


class _Writer:
    def __init__(self):
        self.b = bytearray(8)  # root uoffset + file identifier

    def align(self, n):
        while len(self.b) % n:
            self.b.append(0)

    def patch(self, loc, target):
        struct.pack_into("<I", self.b, loc, target - loc)

    def table(self, fields):
        """fields: list per field index of None | ("i8"|"u8"|"i32"|"u32", value) | ("off", fn) where
        fn() writes the child after this table and returns its position.  Returns the table position."""
        self.align(4)
        sizes = {"i8": 1, "u8": 1, "i32": 4, "u32": 4, "off": 4}
        fmts = {"i8": "<b", "u8": "<B", "i32": "<i", "u32": "<I"}
        # inline layout: soffset, then 4-byte fields, then 1-byte fields
        order = sorted([i for i, f in enumerate(fields) if f is not None], key=lambda i: -sizes[fields[i][0]])
        offs, cur = {}, 4
        for i in order:
            offs[i] = cur
            cur += sizes[fields[i][0]]
        tsize = (cur + 3) & ~3
        vt = struct.pack("<HH", 4 + 2 * len(fields), tsize) + b"".join(struct.pack("<H", offs.get(i, 0)) for i in range(len(fields)))
        if len(vt) % 4:
            vt += b"\0\0"
        vpos = len(self.b)
        self.b += vt
        tpos = len(self.b)
        self.b += bytes(tsize)
        struct.pack_into("<i", self.b, tpos, tpos - vpos)
        pending = []
        for i in order:
            kind, v = fields[i]
            if kind == "off":
                pending.append((tpos + offs[i], v))
            else:
                struct.pack_into(fmts[kind], self.b, tpos + offs[i], v)
        for loc, fn in pending:
            self.patch(loc, fn())
        return tpos

    def vec_scalars(self, fmt, values):
        self.align(4)
        pos = len(self.b)
        self.b += struct.pack("<I", len(values)) + b"".join(struct.pack(fmt, v) for v in values)
        return pos

    def vec_bytes(self, raw):
        self.align(4)
        pos = len(self.b)
        self.b += struct.pack("<I", len(raw)) + raw
        return pos

    def string(self, s):
        self.align(4)
        pos = len(self.b)
        raw = s.encode()
        self.b += struct.pack("<I", len(raw)) + raw + b"\0"
        return pos

    def vec_tables(self, fns):
        self.align(4)
        pos = len(self.b)
        self.b += struct.pack("<I", len(fns)) + bytes(4 * len(fns))
        for i, fn in enumerate(fns):
            self.patch(pos + 4 + 4 * i, fn())
        return pos


def write_mlp_tflite(layers, value_head, policy_head, with_bias=True):
    """layers: 12 x (W [out, in], b [out]) already folded; heads likewise.  Returns bytes."""
    w = _Writer()
    tensors, buffers, ops = [], [b""], []  # buffer 0 is the empty sentinel

    def add_tensor(name, shape, data=None):
        bidx = 0
        if data is not None:
            buffers.append(np.ascontiguousarray(data, "<f4").tobytes())
            bidx = len(buffers) - 1
        tensors.append((name, list(shape), bidx))
        return len(tensors) - 1

    cur = add_tensor("serving_default_input_1:0", [1, layers[0][0].shape[1]])
    inp = cur
    for i, (W, b) in enumerate(layers):
        wi = add_tensor("dense_%d/kernel" % i, W.shape, W)
        bi = add_tensor("dense_%d/bias" % i, b.shape, b) if with_bias else -1
        out = add_tensor("dense_%d/relu" % i, [1, W.shape[0]])
        ops.append((0, [cur, wi, bi], [out], 1))
        cur = out
    outs = []
    for j, ((W, b), code) in enumerate(((value_head, 1), (policy_head, 2))):
        wi = add_tensor("head_%d/kernel" % j, W.shape, W)
        bi = add_tensor("head_%d/bias" % j, b.shape, b) if with_bias else -1
        pre = add_tensor("head_%d/pre" % j, [1, W.shape[0]])
        ops.append((0, [cur, wi, bi], [pre], 0))
        out = add_tensor("StatefulPartitionedCall:%d" % j, [1, W.shape[0]])
        ops.append((code, [pre], [out], 0))
        outs.append(out)

    def tensor_fn(t):
        name, shape, bidx = t
        return lambda: w.table([("off", lambda: w.vec_scalars("<i", shape)), ("i8", 0), ("u32", bidx),
                                ("off", lambda: w.string(name))])

    def op_fn(o):
        opcode_index, ins, outs_, act = o
        fields = [("u32", opcode_index), ("off", lambda: w.vec_scalars("<i", ins)),
                  ("off", lambda: w.vec_scalars("<i", outs_))]
        if opcode_index == 0:
            fields += [("u8", 8), ("off", lambda: w.table([("i8", act)]))]  # FullyConnectedOptions
        return lambda: w.table(fields)

    def subgraph():
        return w.table([("off", lambda: w.vec_tables([tensor_fn(t) for t in tensors])),
                        ("off", lambda: w.vec_scalars("<i", [inp])),
                        ("off", lambda: w.vec_scalars("<i", list(reversed(outs)))),  # policy first, as the reference's files
                        ("off", lambda: w.vec_tables([op_fn(o) for o in ops])),
                        ("off", lambda: w.string("main"))])

    def opcode_fn(code):
        return lambda: w.table([("i8", code), None, ("i32", 1), ("i32", code)])

    def buffer_fn(raw):
        return lambda: w.table([("off", lambda: w.vec_bytes(raw))] if raw else [None])

    root = w.table([("u32", 3),
                    ("off", lambda: w.vec_tables([opcode_fn(c) for c in (9, 28, 25)])),
                    ("off", lambda: w.vec_tables([subgraph])),
                    ("off", lambda: w.string("synthetic test model")),
                    ("off", lambda: w.vec_tables([buffer_fn(r) for r in buffers]))])
    struct.pack_into("<I", w.b, 0, root)
    w.b[4:8] = b"TFL3"
    return bytes(w.b)


def fold_keras_mlp(weights):
    """(layers, value_head, policy_head) as the TFLite converter stores the Keras-order flat
    weights of nets.py: BatchNormalization l folded into layer l + 1 / the heads (float64 fold)."""
    from corintho_ai_amd import nets

    w = np.asarray(weights, np.float32)
    p, fan_in = 0, nets.GAME_STATE_SIZE
    a_prev = c_prev = None
    layers = []

    def fold(K, b):
        K = K.astype(np.float64)
        b = b.astype(np.float64)
        if a_prev is not None:
            b = c_prev @ K + b
            K = a_prev[:, None] * K
        return np.ascontiguousarray(K.T.astype(np.float32)), b.astype(np.float32)

    for _ in range(12):
        K = w[p:p + fan_in * 100].reshape(fan_in, 100)
        p += fan_in * 100
        b, ga, be, mu, va = (w[p + 100 * i:p + 100 * (i + 1)] for i in range(5))
        p += 500
        layers.append(fold(K, b))
        a_prev = ga.astype(np.float64) / np.sqrt(va.astype(np.float64) + nets.BN_EPS)
        c_prev = be.astype(np.float64) - mu.astype(np.float64) * a_prev
        fan_in = 100
    Kv, bv = w[p:p + 100].reshape(100, 1), w[p + 100:p + 101]
    p += 101
    Kp, bp = w[p:p + 9600].reshape(100, 96), w[p + 9600:p + 9696]
    return layers, fold(Kv, bv), fold(Kp, bp)
