"""Precision of the network kernels against a float64 evaluation of the same float32 weights.

The reference evaluates its network in float32 (Keras / TFLite, main.pyx:70-83).  The engine has
three arithmetic widths per architecture:
  * fp32 MFMA (kinds 1, 2): a k-ordered float32 fma chain -- float32 by construction;
  * "bf16x6" (kinds 5, 6): both operands of every product as three bf16 terms whose sum IS the
    float32 value, six MFMA products (everything above 2^-24 of a product), fp32 accumulation --
    claimed float32-equivalent;
  * "bf16x3" (kinds 3, 4): two terms, three products: 16 significand bits, narrower than float32;
  * "f16x3" (kinds 8, 9): two fp16 terms per operand -- 22 significand bits -- three products on the fp16 matrix pipe
    (which keeps fp16 subnormals, tools/exp/f16_denorm.hip): the MFMA work of bf16x3 with the error of the
    float32-wide kinds -- claimed float32-class: within three times the fp32-MFMA kernel's own error.
The claim of the second line is tested here: on random-init weights, on weights with BatchNorm
noise and on trained-scale synthetic weights (kernels x 4, BatchNorm variances 0.1 .. 10, logits
several units wide), the bf16x6 kernels' error against float64 is no larger than twice the fp32-MFMA
kernel's own error (it is about equal or smaller: one accumulator rounding per 16 products instead
of one per product), while bf16x3 sits one to two orders of magnitude above both.
"""
import numpy as np
import pytest

from corintho_ai_amd import (NET_MLP12X100, NET_MLP12X100_H3, NET_MLP12X100_X3, NET_MLP12X100_X6, NET_RESCNN4, NET_RESCNN4_H3,
                             NET_RESCNN4_X3, NET_RESCNN4_X6, nets)
from oracle import oracle as O
from tests import harness as H
from tests import ref_nets
from tests.engines import make_trainer

pytestmark = pytest.mark.gpu


def _states(n, seed):
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 70), np.float32)
    s[:, :64] = rng.integers(0, 2, (n, 64))
    s[:, 64:] = rng.integers(0, 5, (n, 6)) * 0.25
    return s


def _errors(t, kind, w, states, want):
    t.set_net(kind, w)
    ev, pr = t.net_forward(states)
    return float(np.max(np.abs(ev.astype(np.float64) - want[0]))), float(np.max(np.abs(pr.astype(np.float64) - want[1])))


CNN_SETS = [("init", lambda: nets.init_rescnn4(0)), ("bn-noise", lambda: nets.init_rescnn4(3, bn_noise=True)),
            ("trained-like-0", lambda: nets.trained_like_rescnn4(0)), ("trained-like-1", lambda: nets.trained_like_rescnn4(1))]
MLP_SETS = [("init", lambda: nets.init_mlp12x100(0)), ("bn-noise", lambda: nets.init_mlp12x100(7, bn_noise=True)),
            ("trained-like-0", lambda: nets.trained_like_mlp12x100(0)), ("trained-like-1", lambda: nets.trained_like_mlp12x100(1))]


def _check(kinds, sets, f64, label, floor):
    t = make_trainer("hip", 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    states = _states(777, 5)
    rows = []
    for name, make in sets:
        w = make()
        want = f64(w, states)
        e32 = _errors(t, kinds[0], w, states, want)
        e6 = _errors(t, kinds[1], w, states, want)
        e3 = _errors(t, kinds[2], w, states, want)
        eh = _errors(t, kinds[3], w, states, want)
        rows.append((name, e32, e6, e3, eh))
        print("%s %-15s |err| vs float64  value: fp32 %.2e  bf16x6 %.2e  f16x3 %.2e  bf16x3 %.2e   policy: fp32 %.2e  bf16x6 %.2e  "
              "f16x3 %.2e  bf16x3 %.2e" % (label, name, e32[0], e6[0], eh[0], e3[0], e32[1], e6[1], eh[1], e3[1]))
        # f16x3: float32-class -- within three times the fp32-MFMA kernel's error (same floor), and far inside the contract
        assert eh[0] <= 3.0 * e32[0] + floor and eh[1] <= 3.0 * e32[1] + floor, (name, eh, e32)
        assert eh[0] < 1e-5 and eh[1] < 1e-5
        # float32-equivalence: no worse than twice the fp32-MFMA kernel (floor: a few float32 ulps of
        # the output, below which the final tanh / softmax roundings decide)
        assert e6[0] <= 2.0 * e32[0] + floor, (name, e6, e32)
        assert e6[1] <= 2.0 * e32[1] + floor, (name, e6, e32)
        # and inside the north-star contract by a wide margin
        assert e6[0] < 1e-5 and e6[1] < 1e-5
        assert e3[0] < 1e-4 and e3[1] < 1e-4
    return rows


def test_rescnn4_bf16x6_is_float32_equivalent():
    _check((NET_RESCNN4, NET_RESCNN4_X6, NET_RESCNN4_X3, NET_RESCNN4_H3), CNN_SETS, ref_nets.rescnn4_forward_f64, "rescnn4", 2.4e-7)


def test_mlp12x100_bf16x6_is_float32_equivalent():
    _check((NET_MLP12X100, NET_MLP12X100_X6, NET_MLP12X100_X3, NET_MLP12X100_H3), MLP_SETS, ref_nets.mlp12x100_forward_f64, "mlp12x100", 2.4e-7)


@pytest.mark.parametrize("kind,make", [(NET_RESCNN4_X6, lambda: nets.trained_like_rescnn4(2)),
                                       (NET_MLP12X100_X6, lambda: nets.trained_like_mlp12x100(2)),
                                       (NET_RESCNN4_H3, lambda: nets.trained_like_rescnn4(2)),
                                       (NET_MLP12X100_H3, lambda: nets.trained_like_mlp12x100(2))],
                         ids=["rescnn4x6", "mlp12x100x6", "rescnn4h3", "mlp12x100h3"])
def test_bf16x6_rows_do_not_depend_on_their_batch(kind, make):
    """SURVEY 8e invariant: a row's outputs are a function of the row only (fixed k order, no
    batch-dependent tiling), for every batch size around the kernels' tile boundaries"""
    t = make_trainer("hip", 512, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    t.set_net(kind, make())
    states = _states(4096, 9)
    ev, pr = t.net_forward(states)
    assert np.all(np.abs(pr.sum(axis=1) - 1) < 1e-5)
    # ... and around 2048 rows, where the residual CNN changes from its four-wave thin-batch kernel (8 positions per
    # workgroup) to the throughput kernel (16)
    # (the two-term kernels change from 16 to 32 positions per workgroup above 4096 rows: the 8192-row reference batch below)
    states8 = _states(8192, 10)
    ev8, pr8 = t.net_forward(states8)
    e1, p1 = t.net_forward(states8[100:4100])
    assert np.array_equal(e1, ev8[100:4100]) and np.array_equal(p1, pr8[100:4100])
    for lo, hi in ((0, 1), (5, 6), (0, 7), (0, 8), (0, 9), (0, 15), (0, 16), (0, 17), (100, 131), (100, 228), (0, 129), (300, 1024),
                   (0, 2047), (0, 2048), (0, 2049), (1000, 3050), (2040, 4096)):
        e1, p1 = t.net_forward(states[lo:hi])
        assert np.array_equal(e1, ev[lo:hi]) and np.array_equal(p1, pr[lo:hi]), (lo, hi)


def test_a_batch_split_between_the_two_f16x3_kernels_is_the_same_rows():
    """A batch that has the GPU to itself (CoNetIO::alone: net_forward, a one-pool trainer) is split on the device: the
    pixel-major kernel takes its whole passes over the chip (32 rows per CU), the small-batch kernel a remainder of up to
    4096 rows STARTING AT A ROW OFFSET (nn_rescnn.hip rcp_small_begin; its thin path below 2048 rows).  Every row must come
    out as it does when its kernel has the batch to itself."""
    pass_rows = 32 * 256  # an MI355X's 256 CUs (on a device with another count the assertions hold all the same, the split falls elsewhere)
    t = make_trainer("hip", (2 * pass_rows + 4096) // 16 + 1, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0, bn_noise=True))
    states = _states(2 * pass_rows + 4096, 11)
    # whole passes only, and a remainder too large for the small kernel: the pixel-major kernel alone
    ev, pr = t.net_forward(states[:2 * pass_rows])
    e1, p1 = t.net_forward(states[:pass_rows + 4097])
    assert np.array_equal(e1, ev[:pass_rows + 4097]) and np.array_equal(p1, pr[:pass_rows + 4097])
    ref_e = np.concatenate([ev, t.net_forward(states[2 * pass_rows:])[0]])  # (the tail: the small kernel from row 0)
    ref_p = np.concatenate([pr, t.net_forward(states[2 * pass_rows:])[1]])
    for n in (pass_rows + 1, pass_rows + 5, pass_rows + 16, pass_rows + 17, pass_rows + 1808, pass_rows + 2047, pass_rows + 2049,
              pass_rows + 4096, 2 * pass_rows + 1, 2 * pass_rows + 333, 2 * pass_rows + 4096):
        e1, p1 = t.net_forward(states[:n])
        assert np.array_equal(e1, ref_e[:n]) and np.array_equal(p1, ref_p[:n]), n
    # ... and shifted, so that the remainder's rows are other rows
    e1, p1 = t.net_forward(states[77:77 + pass_rows + 1000])
    assert np.array_equal(e1, ref_e[77:77 + pass_rows + 1000]) and np.array_equal(p1, ref_p[77:77 + pass_rows + 1000])


@pytest.mark.parametrize("kind,make", [(NET_RESCNN4_X6, lambda: nets.init_rescnn4(0, bn_noise=True)),
                                       (NET_MLP12X100_X6, lambda: nets.init_mlp12x100(2, bn_noise=True)),
                                       (NET_RESCNN4_H3, lambda: nets.init_rescnn4(0, bn_noise=True)),
                                       (NET_MLP12X100_H3, lambda: nets.init_mlp12x100(2, bn_noise=True))],
                         ids=["rescnn4x6", "mlp12x100x6", "rescnn4h3", "mlp12x100h3"])
def test_fused_bf16x6_generation_replays_on_the_oracle(kind, make):
    G, S_, spe = 24, 40, 8
    f = make_trainer("hip", G, "", 33, S_, spe, 1.0, 0.25, 0, 1, False, stagger=False)
    f.set_net(kind, make())
    assert f.run()
    o = O.Trainer(G, seed=33, max_searches=S_, searches_per_eval=spe)
    o.set_stagger(False)
    H.play_generation(o, G, spe, lambda s: f.net_forward(s))
    for x, y in zip(H.get_samples(f), H.get_samples(o)):
        assert x.tobytes() == y.tobytes()
    assert f.score() == o.score()


# ---------------------------------------------------------------------------------------------------------------
# Range guard of the f16x3 kinds (nn.h CoNet::range_exceeded): an operand beyond fp16's largest finite value would
# turn into infinity, the evaluation into NaN and the tree would run on NaN priors without a word.  Weights are
# refused when the network is set; activations raise a device flag that ends the run with an error naming the x6 kind.
def _scaled(kind, what):
    """adversarial weight sets: `weight` = one weight beyond fp16's range; `activation` = every weight in range, the
    first layer scaled so that its outputs are not"""
    if kind in (NET_RESCNN4_H3, NET_RESCNN4_X6):
        w = nets.init_rescnn4(0, bn_noise=True)
        n_stem = 3 * 3 * nets.RES_CIN * nets.RES_C
        if what == "weight":
            w[n_stem + 6 * nets.RES_C + 5] = 1.0e5  # a kernel weight of the first block's first convolution
        else:
            w[:n_stem] *= 5.0e5  # weights up to 4.7e4, stem outputs ~ 1e5 and beyond
    else:
        w = nets.init_mlp12x100(2, bn_noise=True)
        if what == "weight":
            w[70 * 100 + 500 + 17] = -7.0e5  # a kernel weight of the second layer (times the BatchNorm scale 0.35 .. 2.1 folded into it)
        else:
            w[:70 * 100] *= 2.0e5  # weights up to 3.8e4, first-layer outputs ~ 1e5 and beyond
    return w


@pytest.mark.parametrize("h3,x6", [(NET_RESCNN4_H3, NET_RESCNN4_X6), (NET_MLP12X100_H3, NET_MLP12X100_X6)], ids=["rescnn4", "mlp12x100"])
def test_f16x3_refuses_what_does_not_fit_fp16(h3, x6):
    t = make_trainer("hip", 64, "", 1, 30, 8, 1.0, 0.25, 0, 1, False, stagger=False)
    states = _states(256, 5)
    # (1) a weight beyond the range: refused at set_net, the message names the way out; the x6 kind takes it
    with pytest.raises(RuntimeError, match="x6"):
        t.set_net(h3, _scaled(h3, "weight"))
    t.set_net(x6, _scaled(h3, "weight"))
    ev, pr = t.net_forward(states)
    assert np.all(np.isfinite(ev)) and np.all(np.isfinite(pr))
    # (2) activations beyond the range: the evaluation ends in an error, never in silent NaN; x6 evaluates the same weights
    t.set_net(h3, _scaled(h3, "activation"))
    with pytest.raises(RuntimeError, match="fp16 range"):
        t.net_forward(states)
    with pytest.raises(RuntimeError, match="fp16 range"):
        t.run()
    t.set_net(x6, _scaled(h3, "activation"))
    ev, pr = t.net_forward(states)
    assert np.all(np.isfinite(ev)) and np.all(np.isfinite(pr))
    # (3) in range: nothing is reported, a generation runs
    t2 = make_trainer("hip", 64, "", 1, 30, 8, 1.0, 0.25, 0, 1, False, stagger=False)
    t2.set_net(h3, nets.init_rescnn4(0, bn_noise=True) if h3 == NET_RESCNN4_H3 else nets.init_mlp12x100(2, bn_noise=True))
    assert t2.run()
