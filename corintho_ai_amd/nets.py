"""Network weight layouts of the fused mode, random initialisers and a plain
numpy float32 restatement used by the numerics tests.

mlp12x100 -- the reference architecture (corintho_ai/python/wrapper.py:256-271):
    Input(70) -> 12 x [Dense(100) -> ReLU -> BatchNormalization] ->
    {Dense(1, tanh), Dense(96, softmax)}
flat float32 layout = Keras `get_weights()` order:
    for l in 0..11: kernel[in_l, 100], bias[100], gamma[100], beta[100],
                    moving_mean[100], moving_variance[100]      (in_0 = 70)
    value  head: kernel[100, 1], bias[1]
    policy head: kernel[100, 96], bias[96]
"""
import numpy as np

GAME_STATE_SIZE = 70
NUM_MOVES = 96
BN_EPS = 1e-3  # keras.layers.BatchNormalization default

MLP_NUM_WEIGHTS = 70 * 100 + 500 + 11 * (100 * 100 + 500) + 100 + 1 + 100 * 96 + 96


def _glorot(rng, fan_in, fan_out, shape=None):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape or (fan_in, fan_out)).astype(np.float32)


def init_mlp12x100(seed=0, bn_noise=False):
    """What Keras gives at generation 0: Glorot-uniform kernels, zero biases,
    gamma 1, beta 0, mean 0, variance 1.  bn_noise=True perturbs
    biases and BatchNorm statistics so tests exercise every term."""
    rng = np.random.default_rng(seed)
    parts = []
    fan_in = GAME_STATE_SIZE
    for _ in range(12):
        parts.append(_glorot(rng, fan_in, 100).ravel())
        if bn_noise:
            parts.append(rng.normal(0, 0.1, 100).astype(np.float32))    # bias
            parts.append(rng.uniform(0.5, 1.5, 100).astype(np.float32))  # gamma
            parts.append(rng.normal(0, 0.1, 100).astype(np.float32))    # beta
            parts.append(rng.normal(0, 0.2, 100).astype(np.float32))    # moving mean
            parts.append(rng.uniform(0.5, 2.0, 100).astype(np.float32))  # moving variance
        else:
            parts += [np.zeros(100, np.float32), np.ones(100, np.float32), np.zeros(100, np.float32),
                      np.zeros(100, np.float32), np.ones(100, np.float32)]
        fan_in = 100
    parts.append(_glorot(rng, 100, 1).ravel())
    parts.append(rng.normal(0, 0.1, 1).astype(np.float32) if bn_noise else np.zeros(1, np.float32))
    parts.append(_glorot(rng, 100, NUM_MOVES).ravel())
    parts.append(rng.normal(0, 0.1, NUM_MOVES).astype(np.float32) if bn_noise else np.zeros(NUM_MOVES, np.float32))
    w = np.concatenate(parts).astype(np.float32)
    assert w.size == MLP_NUM_WEIGHTS
    return w


def trained_like_mlp12x100(seed=0):
    """Synthetic weights with the statistics of a trained checkpoint rather than of an initialiser:
    kernels four times the Glorot scale, BatchNorm variances over two decades (0.1 .. 10), non-zero
    biases / means -- logits several units wide, the regime in which a narrow product shows."""
    rng = np.random.default_rng(seed)
    w = init_mlp12x100(seed, bn_noise=True)
    p = 0
    fan_in = GAME_STATE_SIZE
    for _ in range(12):
        w[p:p + fan_in * 100] *= 4.0
        p += fan_in * 100
        w[p + 400:p + 500] = np.exp(rng.uniform(np.log(0.1), np.log(10.0), 100)).astype(np.float32)  # moving variance
        # a trained BatchNorm normalises what it sees: keep the layer's output at unit scale
        w[p + 100:p + 200] = (rng.uniform(0.5, 1.5, 100) * np.sqrt(w[p + 400:p + 500]) / 4.0).astype(np.float32)
        p += 500
        fan_in = 100
    w[p:p + 100] *= 4.0                 # value head
    w[p + 101:p + 101 + 9600] *= 16.0    # policy head: logits a few units wide (entropy ~2.4 nats, as a trained policy)
    return w


# --------------------------------------------------------------------------- rescnn4
# The north-star network (BASELINE.json north_star; it has no counterpart in the
# reference, whose network is the MLP above -- SURVEY "two facts").  Specification:
#   input   the 70-float state as a 4x4 image with 10 channels, NHWC, pixel p = row*4+col:
#           channels 0..3 = the four board bits of the cell (state[p*4 + k]),
#           channels 4..9 = the six reserve counters state[64..69], broadcast over the board
#   stem    Conv3x3(10 -> 64, pad 1) + bias -> BatchNorm -> ReLU
#   4 x     residual block: Conv3x3(64->64)+bias -> BN -> ReLU -> Conv3x3(64->64)+bias -> BN
#           -> add block input -> ReLU
#   policy  Conv1x1(64 -> 4)+bias -> BN -> ReLU -> flatten (pixel*4 + ch) -> Dense(64 -> 96) -> softmax
#   value   Conv1x1(64 -> 2)+bias -> BN -> ReLU -> flatten (pixel*2 + ch) -> Dense(32 -> 64) -> ReLU
#           -> Dense(64 -> 1) -> tanh
# BatchNorm is the inference affine with epsilon 1e-3.  Flat float32 layout, in order:
#   stem: kernel[3,3,10,64] (HWIO), bias[64], gamma, beta, mean, var [64 each]
#   blocks b = 0..3: conv1 (kernel[3,3,64,64], bias, gamma, beta, mean, var), conv2 (same)
#   policy: kernel[64,4], bias[4], gamma, beta, mean, var [4 each], dense kernel[64,96], bias[96]
#   value:  kernel[64,2], bias[2], gamma, beta, mean, var [2 each], dense1 kernel[32,64], bias[64],
#           dense2 kernel[64,1], bias[1]
RES_C = 64
RES_BLOCKS = 4
RES_CIN = 10


def _rescnn4_shapes():
    sh = [("stem_k", (3, 3, RES_CIN, RES_C)), ("stem_b", (RES_C,))] + [("stem_bn%d" % i, (RES_C,)) for i in range(4)]
    for b in range(RES_BLOCKS):
        for c in (1, 2):
            sh += [("b%d_c%d_k" % (b, c), (3, 3, RES_C, RES_C)), ("b%d_c%d_b" % (b, c), (RES_C,))]
            sh += [("b%d_c%d_bn%d" % (b, c, i), (RES_C,)) for i in range(4)]
    sh += [("p_k", (RES_C, 4)), ("p_b", (4,))] + [("p_bn%d" % i, (4,)) for i in range(4)]
    sh += [("p_dk", (64, NUM_MOVES)), ("p_db", (NUM_MOVES,))]
    sh += [("v_k", (RES_C, 2)), ("v_b", (2,))] + [("v_bn%d" % i, (2,)) for i in range(4)]
    sh += [("v_d1k", (32, 64)), ("v_d1b", (64,)), ("v_d2k", (64, 1)), ("v_d2b", (1,))]
    return sh


RESCNN4_NUM_WEIGHTS = sum(int(np.prod(s)) for _, s in _rescnn4_shapes())


def rescnn4_flop_per_row():
    conv = 2 * 16 * 9 * (RES_CIN * RES_C + 2 * RES_BLOCKS * RES_C * RES_C)
    heads = 2 * 16 * RES_C * 6 + 2 * 64 * NUM_MOVES + 2 * 32 * 64 + 2 * 64
    return float(conv + heads)


def rescnn4_useful_flop_per_row():
    """the same count without the products of a 3x3 tap with zero padding: on the 4x4 board 100 of the 144
    (pixel, tap) pairs lie inside it (corners 4 taps, edges 6, interior 9).  The kernels multiply the zeros too."""
    conv = 2 * 16 * 9 * (RES_CIN * RES_C + 2 * RES_BLOCKS * RES_C * RES_C) * 100.0 / 144.0
    heads = 2 * 16 * RES_C * 6 + 2 * 64 * NUM_MOVES + 2 * 32 * 64 + 2 * 64
    return float(conv + heads)


def init_rescnn4(seed=0, bn_noise=False):
    rng = np.random.default_rng(seed)
    parts = []
    for name, shape in _rescnn4_shapes():
        if name.endswith("_k") or name.endswith("k") and not name.endswith("_b"):
            if len(shape) == 4:
                fan_in, fan_out = shape[0] * shape[1] * shape[2], shape[0] * shape[1] * shape[3]
            else:
                fan_in, fan_out = shape
            parts.append(_glorot(rng, fan_in, fan_out, shape).ravel())
        elif "_bn" in name:
            i = int(name[-1])
            if bn_noise:
                v = [rng.uniform(0.5, 1.5, shape), rng.normal(0, 0.1, shape), rng.normal(0, 0.2, shape),
                     rng.uniform(0.5, 2.0, shape)][i]
            else:
                v = [np.ones(shape), np.zeros(shape), np.zeros(shape), np.ones(shape)][i]
            parts.append(np.asarray(v, np.float32).ravel())
        else:  # biases
            parts.append((rng.normal(0, 0.1, shape) if bn_noise else np.zeros(shape)).astype(np.float32).ravel())
    w = np.concatenate(parts).astype(np.float32)
    assert w.size == RESCNN4_NUM_WEIGHTS
    return w


def rescnn4_unpack(weights):
    w = np.asarray(weights, np.float32)
    out, p = {}, 0
    for name, shape in _rescnn4_shapes():
        n = int(np.prod(shape))
        out[name] = w[p:p + n].reshape(shape)
        p += n
    assert p == w.size
    return out


def rescnn4_input_planes(states):
    """[n,70] -> NHWC [n,4,4,10]"""
    s = np.asarray(states, np.float32)[:, :GAME_STATE_SIZE]
    n = s.shape[0]
    x = np.zeros((n, 16, RES_CIN), np.float32)
    x[:, :, :4] = s[:, :64].reshape(n, 16, 4)
    x[:, :, 4:] = s[:, None, 64:70]
    return x.reshape(n, 4, 4, RES_CIN)


def trained_like_rescnn4(seed=0):
    """rescnn4 weights with trained-checkpoint statistics (see trained_like_mlp12x100): kernels x 4,
    BatchNorm variances 0.1 .. 10 with gammas that keep every layer at unit scale."""
    rng = np.random.default_rng(seed)
    w = init_rescnn4(seed, bn_noise=True)
    p = 0
    for name, shape in _rescnn4_shapes():
        n = int(np.prod(shape))
        if name.endswith("k"):
            w[p:p + n] *= 1.0 if name in ("v_d1k", "v_d2k") else 4.0  # (policy entropy ~2.5 nats, value unsaturated)
        elif name.endswith("_bn3"):
            w[p:p + n] = np.exp(rng.uniform(np.log(0.1), np.log(10.0), n)).astype(np.float32)
            ga = p - 3 * n  # gamma sits three arrays before the variance
            w[ga:ga + n] = (rng.uniform(0.5, 1.5, n) * np.sqrt(w[p:p + n]) / 4.0).astype(np.float32)
        p += n
    return w
