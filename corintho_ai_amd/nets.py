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


def mlp12x100_forward_np(weights, states):
    """float32 numpy restatement (Dense -> ReLU -> BN affine; tanh / softmax heads)."""
    w = np.asarray(weights, np.float32)
    x = np.asarray(states, np.float32)[:, :GAME_STATE_SIZE]
    p = 0
    fan_in = GAME_STATE_SIZE
    for _ in range(12):
        K = w[p:p + fan_in * 100].reshape(fan_in, 100)
        p += fan_in * 100
        b, ga, be, mu, va = (w[p + 100 * i:p + 100 * (i + 1)] for i in range(5))
        p += 500
        x = np.maximum(x @ K + b, 0.0).astype(np.float32)
        a = (ga.astype(np.float64) / np.sqrt(va.astype(np.float64) + BN_EPS)).astype(np.float32)
        c = (be.astype(np.float64) - mu.astype(np.float64) * a.astype(np.float64)).astype(np.float32)
        x = (a * x + c).astype(np.float32)
        fan_in = 100
    Kv = w[p:p + 100].reshape(100, 1)
    p += 100
    bv = w[p:p + 1]
    p += 1
    Kp = w[p:p + 9600].reshape(100, 96)
    p += 9600
    bp = w[p:p + 96]
    p += 96
    assert p == w.size
    v = np.tanh((x @ Kv + bv).astype(np.float32)).astype(np.float32)[:, 0]
    lg = (x @ Kp + bp).astype(np.float32)
    lg = lg - lg.max(axis=1, keepdims=True)
    e = np.exp(lg).astype(np.float32)
    return v, (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
