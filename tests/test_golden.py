"""The committed golden vectors (tests/golden/*.npz, written by tools/gen_golden.py) replayed
on every engine: the CPU oracle (so that a change to the oracle cannot pass unnoticed), the
emulation build of the kernel source, and -- with -m gpu -- the MI355X build through the C ABI.
Integer / bit work is compared bit for bit; network outputs within 1e-4 (float32 contract)."""
import os

import numpy as np
import pytest

from corintho_ai_amd import _lib
from tests.engines import ENGINES, cdll, make_trainer
from tools import gen_golden as GG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = ["oracle"] + ENGINES


def load(name):
    with np.load(os.path.join(GOLD, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def test_oracle_reproduces_rules_corpus():
    got = GG.rules_corpus()
    want = load("rules_corpus")
    assert sorted(got) == sorted(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("engine", ENGINES)
def test_rules_corpus(engine):
    """F1: legal masks, is_lines, network input rows and doMove results of 2.9k positions"""
    L = cdll(engine)
    z = load("rules_corpus")
    n = z["boards"].shape[0]
    b, m = z["boards"].copy(), z["metas"].copy()
    out = np.zeros((n, 3), np.uint32)
    ln = np.zeros(n, np.int32)
    _lib.check(L, L.ca_rules_legal_moves(0, b.ctypes.data_as(_lib.u64p), m.ctypes.data_as(_lib.u32p), n,
                                         out.ctypes.data_as(_lib.u32p), ln.ctypes.data_as(_lib.i32p)))
    assert np.array_equal(out, z["masks"])
    assert np.array_equal(ln.astype(np.uint8), z["is_lines"])
    # states of the positions (move -1 = write the state only) ...
    st = np.zeros((n, 70), np.float32)
    mv = np.full(n, -1, np.int32)
    _lib.check(L, L.ca_rules_do_move(0, b.ctypes.data_as(_lib.u64p), m.ctypes.data_as(_lib.u32p),
                                     mv.ctypes.data_as(_lib.i32p), n, st.ctypes.data_as(_lib.f32p)))
    assert np.array_equal(np.packbits((st[:, :64] != 0).astype(np.uint8), axis=1), z["states_packed"])
    assert np.all((st[:, :64] == 0) | (st[:, :64] == 1))
    assert np.array_equal(st[:, 64:], z["state_scalars"])
    # ... and the recorded move applied to each
    mv = z["moves"].copy()
    _lib.check(L, L.ca_rules_do_move(0, b.ctypes.data_as(_lib.u64p), m.ctypes.data_as(_lib.u32p),
                                     mv.ctypes.data_as(_lib.i32p), n, st.ctypes.data_as(_lib.f32p)))
    assert np.array_equal(b, z["boards_after"])
    assert np.array_equal(m, z["metas_after"])
    # the search's four-positions-per-wavefront rule layer (round 5): the same masks, the same moves
    for mv, wb, wm, wmask in ((np.full(n, -1, np.int32), z["boards"], z["metas"], z["masks"]),
                              (z["moves"].copy(), z["boards_after"], z["metas_after"], None)):
        b, m = z["boards"].copy(), z["metas"].copy()
        out = np.zeros((n, 3), np.uint32)
        _lib.check(L, L.ca_rules_rows(0, b.ctypes.data_as(_lib.u64p), m.ctypes.data_as(_lib.u32p),
                                      mv.ctypes.data_as(_lib.i32p), n, out.ctypes.data_as(_lib.u32p)))
        assert np.array_equal(b, wb) and np.array_equal(m, wm)
        if wmask is not None:
            assert np.array_equal(out, wmask)
        else:  # the masks of the positions behind the moves: against the one-position-per-wavefront layer
            ref = np.zeros((n, 3), np.uint32)
            ln2 = np.zeros(n, np.int32)
            _lib.check(L, L.ca_rules_legal_moves(0, wb.copy().ctypes.data_as(_lib.u64p), wm.copy().ctypes.data_as(_lib.u32p), n,
                                                 ref.ctypes.data_as(_lib.u32p), ln2.ctypes.data_as(_lib.i32p)))
            assert np.array_equal(out, ref)


def _factory(engine):
    if engine == "oracle":
        return None

    def make(G, seed, sims, spe, c_puct, eps, testing, stagger):
        return make_trainer(engine, G, "", seed, sims, spe, c_puct, eps, 0, 1, testing, trace=True, stagger=stagger)

    return make


@pytest.mark.parametrize("engine", ALL)
@pytest.mark.parametrize("case", GG.SELFPLAY_CASES, ids=lambda c: c[0])
def test_selfplay_golden(engine, case):
    """F2-F4: iterations, request counts and rows of every iteration, per-ply traces, results,
    sample tensors, score and mate length of whole generations"""
    name, *cfg = case
    want = load(name)
    got = GG.selfplay_case(*cfg, trainer_factory=_factory(engine))
    assert sorted(got) == sorted(want)
    for k in want:
        a, b = np.asarray(got[k]), want[k]
        if a.dtype.kind == "f":
            assert a.tobytes() == b.tobytes(), k
        else:
            assert np.array_equal(a, b), k


def _net_cases():
    """(engine, vector name, network kind): the MLP on every engine, the residual CNN kernels
    (which have no emulation build) on the GPU"""
    from corintho_ai_amd import NET_MLP12X100, NET_MLP12X100_X3, NET_RESCNN4, NET_RESCNN4_X3

    gpu = pytest.mark.gpu
    out = []
    for name in ("mlp_seed0", "mlp_seed1_noise"):
        out.append(pytest.param("emu", name, NET_MLP12X100, id="emu-" + name))
        out.append(pytest.param("hip", name, NET_MLP12X100, id="hip-" + name, marks=gpu))
        out.append(pytest.param("hip", name, NET_MLP12X100_X3, id="hip-bf16x3-" + name, marks=gpu))
    for name in ("rescnn4_seed0", "rescnn4_seed3_noise"):
        out.append(pytest.param("hip", name, NET_RESCNN4, id="hip-fp32-" + name, marks=gpu))
        out.append(pytest.param("hip", name, NET_RESCNN4_X3, id="hip-bf16x3-" + name, marks=gpu))
    return out


def test_restatements_reproduce_net_vectors():
    z = load("net_vectors")
    got = GG.net_vectors()
    assert np.array_equal(got["states"], z["states"])
    for k in z:
        if k.endswith("_sha256"):
            assert str(got[k]) == str(z[k]), k  # numpy's generator must give the same weights everywhere
        elif k != "states":
            assert np.max(np.abs(got[k] - z[k])) < 2e-6, k  # BLAS summation order may differ between hosts


@pytest.mark.parametrize("engine,name,kind", _net_cases())
def test_net_vectors(engine, name, kind):
    """network logits of states met in play vs the committed float32 vectors (<= 1e-4)"""
    z = load("net_vectors")
    t = make_trainer(engine, 64, "", 1, 50, 16, 1.0, 0.25, 0, 1, False)
    t.set_net(kind, GG.NET_INITS[name]())
    ev, pr = t.net_forward(z["states"])
    assert np.max(np.abs(ev - z[name + "_value"])) < 1e-4
    assert np.max(np.abs(pr - z[name + "_policy"])) < 1e-4
