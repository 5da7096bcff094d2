"""Weight import from the reference's TFLite checkpoints.

The reference rates and serves its networks as TFLite flatbuffers
(corintho_ai/rating/tflite_models/model_*.tflite, loaded by rating/tourney.pyx:139-154 and
docker/choose_move.pyx:118-133 with `tflite_runtime`).  The converter has folded every
BatchNormalization into the Dense layer that follows it, so a file holds the network of
wrapper.py:256-271 as a chain of FULLY_CONNECTED operators:

    h0 = relu(W0 x + b0);  h_l = relu(W_l' h_{l-1} + b_l')  (l = 1..11)
    value = tanh(Wv' h11 + bv');  policy = softmax(Wp' h11 + bp')

with W_l' = W_l diag(a_{l-1}), b_l' = W_l c_{l-1} + b_l.  `mlp12x100_from_tflite` returns that
network in the engine's flat Keras-order layout (nets.py) with the folded kernels and an
identity BatchNormalization (gamma 1, beta 0, mean 0, variance 1 - eps), which the fused
kernel (csrc/nn_mlp.hip) evaluates as exactly the same function.

No TFLite runtime is needed (none is installed here): the file is read with a ~60-line
FlatBuffers table reader that knows the handful of schema fields involved
(tensorflow/lite/schema/schema.fbs: Model, SubGraph, Tensor, Operator, OperatorCode, Buffer).
"""
import struct

import numpy as np

from . import nets

# BuiltinOperator codes used by the reference's models
OP_FULLY_CONNECTED, OP_SOFTMAX, OP_TANH = 9, 25, 28
ACT_NONE, ACT_RELU = 0, 1
TENSOR_FLOAT32 = 0


class TFLiteFormatError(ValueError):
    pass


class _Table:
    """FlatBuffers table: `pos` points at the table, its vtable is at pos - int32(pos)"""

    def __init__(self, data, pos):
        self.d, self.pos = data, pos
        self.vt = pos - struct.unpack_from("<i", data, pos)[0]
        self.vsz = struct.unpack_from("<H", data, self.vt)[0]

    def _off(self, field):
        o = 4 + 2 * field
        return struct.unpack_from("<H", self.d, self.vt + o)[0] if o < self.vsz else 0

    def scalar(self, field, fmt, default=0):
        o = self._off(field)
        return struct.unpack_from(fmt, self.d, self.pos + o)[0] if o else default

    def _indirect(self, field):
        o = self._off(field)
        if not o:
            return None
        p = self.pos + o
        return p + struct.unpack_from("<I", self.d, p)[0]

    def table(self, field):
        p = self._indirect(field)
        return _Table(self.d, p) if p is not None else None

    def vector(self, field):
        """(count, offset of element 0)"""
        p = self._indirect(field)
        if p is None:
            return 0, 0
        return struct.unpack_from("<I", self.d, p)[0], p + 4

    def string(self, field):
        n, p = self.vector(field)
        return bytes(self.d[p:p + n]).decode("utf-8", "replace") if n else ""

    def tables(self, field):
        n, p = self.vector(field)
        return [_Table(self.d, p + 4 * i + struct.unpack_from("<I", self.d, p + 4 * i)[0]) for i in range(n)]

    def ints(self, field):
        n, p = self.vector(field)
        return list(struct.unpack_from("<%di" % n, self.d, p)) if n else []


def read_tflite(path_or_bytes):
    """-> {"tensors": [{name, shape, type, data (np.float32 array or None)}], "ops": [{code, inputs,
    outputs, activation}], "inputs": [...], "outputs": [...]} of the first subgraph"""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    if len(data) < 8 or data[4:8] != b"TFL3":
        raise TFLiteFormatError("not a TFLite flatbuffer (file identifier TFL3 missing)")
    root = _Table(data, struct.unpack_from("<I", data, 0)[0])
    # OperatorCode: deprecated_builtin_code (field 0, int8) for codes < 127, builtin_code (field 3)
    codes = [max(c.scalar(0, "<b"), c.scalar(3, "<i")) for c in root.tables(1)]
    subgraphs = root.tables(2)
    if not subgraphs:
        raise TFLiteFormatError("model without a subgraph")
    sg = subgraphs[0]
    buffers = root.tables(4)
    tensors = []
    for t in sg.tables(0):
        shape = t.ints(0)
        ttype = t.scalar(1, "<b")
        bidx = t.scalar(2, "<I")
        arr = None
        if bidx < len(buffers):
            n, p = buffers[bidx].vector(0)
            if n and ttype == TENSOR_FLOAT32:
                arr = np.frombuffer(data, dtype="<f4", count=n // 4, offset=p).reshape(shape).astype(np.float32)
        tensors.append({"name": t.string(3), "shape": shape, "type": ttype, "data": arr})
    ops = []
    for op in sg.tables(3):
        code = codes[op.scalar(0, "<I")]
        act = ACT_NONE
        if code == OP_FULLY_CONNECTED:
            opt = op.table(4)  # FullyConnectedOptions: fused_activation_function = field 0 (int8)
            act = opt.scalar(0, "<b") if opt is not None else ACT_NONE
        ops.append({"code": code, "inputs": op.ints(1), "outputs": op.ints(2), "activation": act})
    return {"tensors": tensors, "ops": ops, "inputs": sg.ints(1), "outputs": sg.ints(2)}


def tflite_forward_np(model, states, dtype=np.float32):
    """Evaluates the operators of `model` (read_tflite) as stored, in numpy at `dtype` (float32 = what the
    TFLite runtime computes; float64 = the yardstick of the precision tests): the check of the import
    (tests) -- FULLY_CONNECTED (+ fused ReLU), TANH, SOFTMAX only.  Returns {output tensor index: array}."""
    T = model["tensors"]
    val = {model["inputs"][0]: np.asarray(states, np.float32)[:, :nets.GAME_STATE_SIZE].astype(dtype)}
    for op in model["ops"]:
        x = val[op["inputs"][0]]
        if op["code"] == OP_FULLY_CONNECTED:
            w = T[op["inputs"][1]]["data"].astype(dtype)  # [out, in]
            y = (x @ w.T).astype(dtype)
            if len(op["inputs"]) > 2 and op["inputs"][2] >= 0:
                y = (y + T[op["inputs"][2]]["data"].astype(dtype).reshape(1, -1)).astype(dtype)
            if op["activation"] == ACT_RELU:
                y = np.maximum(y, 0.0)
            elif op["activation"] != ACT_NONE:
                raise TFLiteFormatError("unsupported fused activation %d" % op["activation"])
        elif op["code"] == OP_TANH:
            y = np.tanh(x).astype(dtype)
        elif op["code"] == OP_SOFTMAX:
            z = x - x.max(axis=1, keepdims=True)
            e = np.exp(z).astype(dtype)
            y = (e / e.sum(axis=1, keepdims=True)).astype(dtype)
        else:
            raise TFLiteFormatError("unsupported operator %d" % op["code"])
        val[op["outputs"][0]] = y
    return {i: val[i] for i in model["outputs"]}


def mlp12x100_from_tflite(path_or_bytes):
    """-> flat float32 weights (nets.MLP_NUM_WEIGHTS) of the network stored in a reference
    TFLite checkpoint.  Raises TFLiteFormatError when the graph is not the 12 x 100 MLP."""
    m = read_tflite(path_or_bytes)
    T = m["tensors"]
    if len(m["inputs"]) != 1 or T[m["inputs"][0]]["shape"][-1] != nets.GAME_STATE_SIZE:
        raise TFLiteFormatError("expected one input of %d floats" % nets.GAME_STATE_SIZE)

    def fc_params(op, n_in, n_out):
        w = T[op["inputs"][1]]["data"]
        if w is None or list(w.shape) != [n_out, n_in]:
            raise TFLiteFormatError("FULLY_CONNECTED weights %s, expected [%d, %d]" %
                                    (None if w is None else list(w.shape), n_out, n_in))
        b = None
        if len(op["inputs"]) > 2 and op["inputs"][2] >= 0:
            b = T[op["inputs"][2]]["data"]
        b = np.zeros(n_out, np.float32) if b is None else b.reshape(-1).astype(np.float32)
        if b.size != n_out:
            raise TFLiteFormatError("bias of %d elements, expected %d" % (b.size, n_out))
        return np.ascontiguousarray(w.T, np.float32), b  # Keras kernel layout [in, out]

    # follow the trunk from the input
    producers_of = {}
    for op in m["ops"]:
        producers_of.setdefault(op["inputs"][0], []).append(op)
    cur = m["inputs"][0]
    parts = []
    n_in = nets.GAME_STATE_SIZE
    identity_bn = [np.ones(100, np.float32), np.zeros(100, np.float32), np.zeros(100, np.float32),
                   np.full(100, 1.0 - nets.BN_EPS, np.float32)]  # gamma, beta, mean, variance: a = 1, c = 0
    for layer in range(12):
        nxt = [op for op in producers_of.get(cur, []) if op["code"] == OP_FULLY_CONNECTED]
        if len(nxt) != 1 or nxt[0]["activation"] != ACT_RELU:
            raise TFLiteFormatError("layer %d: expected one FULLY_CONNECTED with fused ReLU" % layer)
        k, b = fc_params(nxt[0], n_in, 100)
        parts += [k.ravel(), b] + identity_bn
        cur = nxt[0]["outputs"][0]
        n_in = 100
    heads = [op for op in producers_of.get(cur, []) if op["code"] == OP_FULLY_CONNECTED]
    value = policy = None
    for op in heads:
        follow = producers_of.get(op["outputs"][0], [])
        if len(follow) == 1 and follow[0]["code"] == OP_TANH and op["activation"] == ACT_NONE:
            value = fc_params(op, 100, 1)
        elif len(follow) == 1 and follow[0]["code"] == OP_SOFTMAX and op["activation"] == ACT_NONE:
            policy = fc_params(op, 100, nets.NUM_MOVES)
    if value is None or policy is None:
        raise TFLiteFormatError("expected a Dense(1) -> TANH head and a Dense(96) -> SOFTMAX head")
    parts += [value[0].ravel(), value[1], policy[0].ravel(), policy[1]]
    w = np.concatenate(parts).astype(np.float32)
    if w.size != nets.MLP_NUM_WEIGHTS:
        raise TFLiteFormatError("unexpected number of weights %d" % w.size)
    return w


def output_roles(model):
    """{'value': tensor index, 'policy': tensor index} of a read_tflite model (the reference
    reads them by position: tourney.pyx:153-154)"""
    roles = {}
    for op in model["ops"]:
        if op["code"] == OP_TANH:
            roles["value"] = op["outputs"][0]
        if op["code"] == OP_SOFTMAX:
            roles["policy"] = op["outputs"][0]
    return roles
