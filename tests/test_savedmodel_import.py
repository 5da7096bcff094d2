"""SURVEY 8f row 3, second format: the reference's Keras SavedModel (corintho_ai/model, loaded at main.pyx:296) read
without TensorFlow (corintho_ai_amd/savedmodel_import.py).  The format reader is tested on bundles written here (a few
dozen lines of the same table format: test infrastructure); tests/golden/savedmodel.npz holds the reference's own
variables as data (tools/gen_savedmodel_golden.py) -- the one fixture of reference weights with BatchNorm UNFOLDED
(gamma, beta, moving statistics; the TFLite checkpoints carry it folded into the next layer) -- and runs through the
network kernels at every arithmetic width."""
import os
import struct

import numpy as np
import pytest

from corintho_ai_amd import NET_MLP12X100, NET_MLP12X100_H3, NET_MLP12X100_X3, NET_MLP12X100_X6, nets
from corintho_ai_amd import savedmodel_import as SI
from oracle import oracle as O
from tests import harness as H
from tests import ref_nets
from tests.engines import ENGINES, make_trainer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "savedmodel.npz")
REF_MODEL = "/root/reference/corintho_ai/model"


# ---- a TensorBundle writer for the tests: sorted table with prefix compression, several data blocks, one shard
def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _block(entries, restart_every=4):
    body, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_every == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        body += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _entry_proto(dtype, shape, offset, size):
    dims = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in shape))
    return b"\x08" + _vi(dtype) + (b"\x12" + _vi(len(dims)) + dims if shape else b"") + b"\x20" + _vi(offset) + b"\x28" + _vi(size) + \
        b"\x35" + b"\0\0\0\0"


def write_bundle(prefix, tensors, per_block=5):
    data, entries = bytearray(), [(b"", b"\x08\x01\x1a\x02\x08\x01")]  # header: one shard, version 1
    for name in sorted(tensors):
        a = np.ascontiguousarray(tensors[name])
        dt = {np.dtype("float32"): 1, np.dtype("int64"): 9}[a.dtype]
        entries.append((name.encode(), _entry_proto(dt, a.shape, len(data), a.nbytes)))
        data += a.tobytes()
    out, index = bytearray(), []
    for i in range(0, len(entries), per_block):
        blk = _block(entries[i:i + per_block])
        index.append((entries[min(i + per_block, len(entries)) - 1][0] + b"\xff", _vi(len(out)) + _vi(len(blk))))
        out += blk + b"\0" + b"\0\0\0\0"
    meta_off = len(out)
    meta = _block([])
    out += meta + b"\0" + b"\0\0\0\0"
    idx_off = len(out)
    idx = _block(index, restart_every=1)
    out += idx + b"\0" + b"\0\0\0\0"
    foot = _vi(meta_off) + _vi(len(meta)) + _vi(idx_off) + _vi(len(idx))
    out += foot + b"\0" * (40 - len(foot)) + struct.pack("<Q", SI.TABLE_MAGIC)
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    with open(prefix + ".index", "wb") as f:
        f.write(out)
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)


def _keras_variables(w):
    """the flat layout -> the variables Keras saves for the network of wrapper.py:256-271"""
    t, off, n_in = {}, 0, 70
    suf = "/.ATTRIBUTES/VARIABLE_VALUE"

    def take(n, shape):
        nonlocal off
        a = w[off:off + n].reshape(shape).copy()
        off += n
        return a

    for layer in range(12):
        t["layer_with_weights-%d/kernel%s" % (2 * layer, suf)] = take(n_in * 100, (n_in, 100))
        t["layer_with_weights-%d/bias%s" % (2 * layer, suf)] = take(100, (100,))
        for name in ("gamma", "beta", "moving_mean", "moving_variance"):
            t["layer_with_weights-%d/%s%s" % (2 * layer + 1, name, suf)] = take(100, (100,))
        n_in = 100
    t["layer_with_weights-24/kernel" + suf] = take(100, (100, 1))
    t["layer_with_weights-24/bias" + suf] = take(1, (1,))
    t["layer_with_weights-25/kernel" + suf] = take(9600, (100, 96))
    t["layer_with_weights-25/bias" + suf] = take(96, (96,))
    t["optimizer/iter" + suf] = np.array(1234, np.int64)  # (a scalar of another dtype, and names that sort between)
    t["keras_api/metrics/0/total" + suf] = np.array(0.5, np.float32)
    return t


def test_reader_round_trips_a_written_bundle(tmp_path):
    w = nets.init_mlp12x100(seed=5, bn_noise=True)
    for per_block in (1, 5, 1000):
        d = tmp_path / ("m%d" % per_block)
        write_bundle(str(d / "variables" / "variables"), _keras_variables(w), per_block)
        t = SI.read_tensor_bundle(str(d / "variables" / "variables"))
        assert t["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"].reshape(-1)[0] == 1234
        assert SI.mlp12x100_from_savedmodel(str(d)).tobytes() == w.tobytes()


def test_reader_refuses_what_it_does_not_understand(tmp_path):
    w = nets.init_mlp12x100(seed=5)
    v = _keras_variables(w)
    d = tmp_path / "m"
    write_bundle(str(d / "variables" / "variables"), v)
    raw = bytearray(open(str(d / "variables" / "variables.index"), "rb").read())
    with pytest.raises(SI.SavedModelFormatError, match="footer"):
        SI.read_index(bytes(raw[:-1]))
    bad = bytearray(raw)
    bad[-8] ^= 1
    with pytest.raises(SI.SavedModelFormatError, match="footer"):
        SI.read_index(bytes(bad))
    # another architecture: a kernel of the wrong shape, a missing layer
    v2 = dict(v)
    v2["layer_with_weights-2/kernel/.ATTRIBUTES/VARIABLE_VALUE"] = np.zeros((100, 50), np.float32)
    write_bundle(str(tmp_path / "b" / "variables" / "variables"), v2)
    with pytest.raises(SI.SavedModelFormatError, match="shape"):
        SI.mlp12x100_from_savedmodel(str(tmp_path / "b"))
    v3 = {k: a for k, a in v.items() if "layer_with_weights-25/" not in k}
    write_bundle(str(tmp_path / "c" / "variables" / "variables"), v3)
    with pytest.raises(SI.SavedModelFormatError, match="no variable"):
        SI.mlp12x100_from_savedmodel(str(tmp_path / "c"))


def load():
    with np.load(GOLDEN) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.skipif(not os.path.isdir(REF_MODEL), reason="reference tree not mounted")
def test_the_fixture_is_the_reference_model():
    """build container only: the committed weights are what the reader finds in the reference's model directory"""
    assert SI.mlp12x100_from_savedmodel(REF_MODEL).tobytes() == load()["weights"].tobytes()


def test_fixture_holds_batchnorm_unfolded():
    z = load()
    s = z["bn_stats"]  # per layer: min / max of gamma, beta, mean, variance
    assert s.shape == (12, 8)
    assert np.all(s[:, 1] - s[:, 0] > 0.1) and np.all(s[:, 7] > 3 * s[:, 6]) and np.all(s[:, 6] > 0)  # nothing like the identity
    v32, p32 = ref_nets.mlp12x100_forward_np(z["weights"], z["states"])
    assert np.max(np.abs(v32 - z["value_f64"])) < 2e-5 and np.max(np.abs(p32 - z["policy_f64"])) < 2e-5
    assert np.allclose(z["policy_f64"].sum(1), 1.0, atol=1e-12) and np.max(z["policy_f64"]) > 0.3  # a trained network: sharp priors


@pytest.mark.parametrize("engine", ENGINES)
def test_fp32_kernel_on_the_saved_model(engine):
    z = load()
    t = make_trainer(engine, 16, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    t.set_net(NET_MLP12X100, z["weights"])
    ev, pr = t.net_forward(z["states"])
    assert np.max(np.abs(ev - z["value_f64"])) < 1e-4 and np.max(np.abs(pr - z["policy_f64"])) < 1e-4


@pytest.mark.gpu
def test_every_width_on_the_saved_model_and_a_generation_on_the_oracle():
    z = load()
    t = make_trainer("hip", 24, "", 5, 60, 16, 1.0, 0.25, 0, 1, False, stagger=False, trace=True)
    err = {}
    for name, kind in (("fp32", NET_MLP12X100), ("bf16x6", NET_MLP12X100_X6), ("f16x3", NET_MLP12X100_H3), ("bf16x3", NET_MLP12X100_X3)):
        t.set_net(kind, z["weights"])
        ev, pr = t.net_forward(z["states"])
        err[name] = (float(np.max(np.abs(ev - z["value_f64"]))), float(np.max(np.abs(pr - z["policy_f64"]))))
        tol = 1e-3 if name == "bf16x3" else 1e-4  # (narrower than float32: a throughput variant, never the default)
        assert err[name][0] < tol and err[name][1] < tol, (name, err[name])
    print("saved model: |err| vs float64 (value, policy): %s" % err)
    assert err["f16x3"][0] <= 3 * err["fp32"][0] + 5e-7 and err["f16x3"][1] <= 3 * err["fp32"][1] + 5e-7
    # ... and a fused generation guided by it (explicit BatchNorm folded on the host for the split kernels) replays on the oracle
    t.set_net(NET_MLP12X100_H3, z["weights"])
    assert t.run()
    o = O.Trainer(24, seed=5, max_searches=60, searches_per_eval=16)
    o.enable_trace()
    o.set_stagger(False)
    H.play_generation(o, 24, 16, lambda s: t.net_forward(s))
    for g in range(24):
        assert np.array_equal(t.trace(g), o.trace(g)), g
    assert all(x.tobytes() == y.tobytes() for x, y in zip(H.get_samples(t), H.get_samples(o)))
