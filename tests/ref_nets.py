"""TEST INFRASTRUCTURE: float32 / float64 restatements of the two networks on the CPU (numpy, torch-CPU convolutions).
The yardsticks of the precision tests, the fixtures' generators and the CPU leg of bench.py -- never imported by the
product package (corintho_ai_amd/nets.py keeps the weight layouts and initialisers only).  Moved here in round 5."""
import numpy as np

from corintho_ai_amd.nets import (BN_EPS, GAME_STATE_SIZE, NUM_MOVES, RES_BLOCKS, rescnn4_input_planes,  # noqa: F401
                                  rescnn4_unpack)


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


def mlp12x100_forward_f64(weights, states):
    """the same function of the float32 weights evaluated in float64 throughout (BatchNorm constants
    included): the yardstick the float32 kernels' errors are measured against"""
    w = np.asarray(weights, np.float32).astype(np.float64)
    x = np.asarray(states, np.float32)[:, :GAME_STATE_SIZE].astype(np.float64)
    p = 0
    fan_in = GAME_STATE_SIZE
    for _ in range(12):
        K = w[p:p + fan_in * 100].reshape(fan_in, 100)
        p += fan_in * 100
        b, ga, be, mu, va = (w[p + 100 * i:p + 100 * (i + 1)] for i in range(5))
        p += 500
        x = np.maximum(x @ K + b, 0.0)
        a = ga / np.sqrt(va + BN_EPS)
        x = a * x + (be - mu * a)
        fan_in = 100
    Kv = w[p:p + 100].reshape(100, 1)
    p += 100
    bv = w[p:p + 1]
    p += 1
    Kp = w[p:p + 9600].reshape(100, 96)
    p += 9600
    bp = w[p:p + 96]
    v = np.tanh(x @ Kv + bv)[:, 0]
    lg = x @ Kp + bp
    lg = lg - lg.max(axis=1, keepdims=True)
    e = np.exp(lg)
    return v, e / e.sum(axis=1, keepdims=True)


def rescnn4_forward_f64(weights, states):
    """the same function of the float32 weights evaluated in float64 throughout"""
    return rescnn4_forward_ref(weights, states, f64=True)


def rescnn4_forward_ref(weights, states, f64=False):
    """float32 restatement with torch CPU convolutions (test infrastructure); f64: float64 throughout."""
    import torch
    import torch.nn.functional as F

    if f64:
        return _rescnn4_forward_f64(weights, states)
    W = rescnn4_unpack(weights)
    x = torch.from_numpy(rescnn4_input_planes(states)).permute(0, 3, 1, 2).contiguous()  # NCHW

    def bn(t, prefix):
        ga, be, mu, va = (torch.from_numpy(W[prefix + "_bn%d" % i].astype(np.float64)) for i in range(4))
        a = (ga / torch.sqrt(va + BN_EPS)).float()
        c = (be - mu * a.double()).float()
        return t * a.view(1, -1, 1, 1) + c.view(1, -1, 1, 1)

    def conv(t, prefix, pad):
        k = torch.from_numpy(W[prefix + "_k"])
        if k.dim() == 2:
            k = k.view(1, 1, *k.shape)
        k = k.permute(3, 2, 0, 1).contiguous()  # HWIO -> OIHW
        return F.conv2d(t, k, torch.from_numpy(W[prefix + "_b"]), padding=pad)

    with torch.no_grad():
        x = torch.relu(bn(conv(x, "stem", 1), "stem"))
        for b in range(RES_BLOCKS):
            y = torch.relu(bn(conv(x, "b%d_c1" % b, 1), "b%d_c1" % b))
            y = bn(conv(y, "b%d_c2" % b, 1), "b%d_c2" % b)
            x = torch.relu(x + y)
        p = torch.relu(bn(conv(x, "p", 0), "p"))            # [n,4,4,4] NCHW
        p = p.permute(0, 2, 3, 1).reshape(p.shape[0], 64)    # pixel*4 + ch
        logits = p @ torch.from_numpy(W["p_dk"]) + torch.from_numpy(W["p_db"])
        probs = torch.softmax(logits, dim=1)
        v = torch.relu(bn(conv(x, "v", 0), "v"))
        v = v.permute(0, 2, 3, 1).reshape(v.shape[0], 32)    # pixel*2 + ch
        v = torch.relu(v @ torch.from_numpy(W["v_d1k"]) + torch.from_numpy(W["v_d1b"]))
        v = torch.tanh(v @ torch.from_numpy(W["v_d2k"]) + torch.from_numpy(W["v_d2b"]))[:, 0]
    return v.numpy().astype(np.float32), probs.numpy().astype(np.float32)


def _rescnn4_forward_f64(weights, states):
    import torch
    import torch.nn.functional as F

    W = {k: torch.from_numpy(v.astype(np.float64)) for k, v in rescnn4_unpack(weights).items()}
    x = torch.from_numpy(rescnn4_input_planes(states).astype(np.float64)).permute(0, 3, 1, 2).contiguous()

    def bn(t, prefix):
        ga, be, mu, va = (W[prefix + "_bn%d" % i] for i in range(4))
        a = ga / torch.sqrt(va + BN_EPS)
        return t * a.view(1, -1, 1, 1) + (be - mu * a).view(1, -1, 1, 1)

    def conv(t, prefix, pad):
        k = W[prefix + "_k"]
        if k.dim() == 2:
            k = k.view(1, 1, *k.shape)
        return F.conv2d(t, k.permute(3, 2, 0, 1).contiguous(), W[prefix + "_b"], padding=pad)

    with torch.no_grad():
        x = torch.relu(bn(conv(x, "stem", 1), "stem"))
        for b in range(RES_BLOCKS):
            y = torch.relu(bn(conv(x, "b%d_c1" % b, 1), "b%d_c1" % b))
            y = bn(conv(y, "b%d_c2" % b, 1), "b%d_c2" % b)
            x = torch.relu(x + y)
        p = torch.relu(bn(conv(x, "p", 0), "p"))
        p = p.permute(0, 2, 3, 1).reshape(p.shape[0], 64)
        probs = torch.softmax(p @ W["p_dk"] + W["p_db"], dim=1)
        v = torch.relu(bn(conv(x, "v", 0), "v"))
        v = v.permute(0, 2, 3, 1).reshape(v.shape[0], 32)
        v = torch.relu(v @ W["v_d1k"] + W["v_d1b"])
        v = torch.tanh(v @ W["v_d2k"] + W["v_d2b"])[:, 0]
    return v.numpy(), probs.numpy()
