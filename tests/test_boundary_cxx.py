"""The source-level boundary EXECUTED: `class Trainer` of corintho_ai_amd/cpp/trainer.{h,cpp} -- the
class the reference's Cython module consumes (main.pyx:17-38) -- driven from C++ by the reference's
own play loop (tests/cxx/trainer_driver.cpp restates main.pyx:123-219), with a stand-in network
coded in C++.  Every request batch, the three sample arrays, score, mate length and the scores
file must equal the CPU oracle's, bit for bit.  Runs against the emulation build of the engine
here and against libcorintho_hip.so on the MI355X (-m gpu).

Second test (build container only): the reference's own main.pyx, with nothing but the `extern
from` path changed, cythonizes and COMPILES against this Trainer and links with the engine library
(it is not imported: keras is absent)."""
import os
import re
import subprocess
import sys
import sysconfig

import numpy as np
import pytest

from oracle import oracle as O
from tests import harness as H
from tests.conftest import REFERENCE
from tests.engines import ENGINES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_for(engine):
    if engine == "emu":
        from tests.emu import emulib

        emulib.load()
        return os.path.join(ROOT, "tests", "emu"), "corintho_emu"
    from corintho_ai_amd import build

    build.build()
    return os.path.join(ROOT, "corintho_ai_amd"), "corintho_hip"


def _build_driver(tmp_path, engine):
    libdir, libname = _lib_for(engine)
    exe = str(tmp_path / ("trainer_driver_" + engine))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cxx", "trainer_driver.cpp"),
                           "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-fopenmp"])
    return exe


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("testing", [0, 1], ids=["train", "arena"])
def test_cpp_trainer_plays_a_generation_equal_to_the_oracle(engine, testing, tmp_path):
    G, seed, S_, spe = 16, 4711, 40, 8
    exe = _build_driver(tmp_path, engine)
    prefix = str(tmp_path / "run")
    os.mkdir(prefix + ".logs")
    os.mkdir(str(tmp_path / "oracle_logs"))
    subprocess.check_call([exe, prefix, str(G), str(seed), str(S_), str(spe), str(testing)])
    # the oracle through the same loop in Python, with the same stand-in network
    o = O.Trainer(G, str(tmp_path / "oracle_logs"), seed, S_, spe, 1.0, 0.25, 2, 1, bool(testing))
    if testing:
        r = H.play_generation(o, G, spe, None, nets_by_player=(lambda s: H.hash_net(s, 1), lambda s: H.hash_net(s, 2)), record=True)
    else:
        r = H.play_generation(o, G, spe, H.hash_net, record=True)
    raw = open(prefix + ".requests.bin", "rb").read()
    pos, got = 0, []
    while pos < len(raw):
        tp, n = np.frombuffer(raw, np.int32, 2, pos)
        pos += 8
        got.append((int(tp), raw[pos:pos + int(n) * 70 * 4]))
        pos += int(n) * 70 * 4
    assert len(got) == len(r["log"])
    for i, (a, b) in enumerate(zip(got, r["log"])):
        assert a[0] == b[0] and a[1] == b[1].tobytes(), "request batch %d" % i
    raw = open(prefix + ".samples.bin", "rb").read()
    ns, iters = np.frombuffer(raw, np.int32, 2, 0)
    score, mate = np.frombuffer(raw, np.float32, 2, 8)
    assert int(iters) == r["iterations"] and int(ns) == o.num_samples()
    assert float(score) == np.float32(o.score()) and float(mate) == np.float32(o.avg_mate_length())
    gs, ev, pr = H.get_samples(o)
    assert raw[16:] == gs.tobytes() + ev.tobytes() + pr.tobytes()
    fo = str(tmp_path / "oracle_scores.txt")
    o.writeScores(fo)
    assert open(prefix + ".scores.txt", "rb").read() == open(fo, "rb").read()
    # the per-game text logs of the first two games
    assert sorted(os.listdir(prefix + ".logs")) == ["game_0.txt", "game_1.txt"]
    for name in ("game_0.txt", "game_1.txt"):
        assert open(os.path.join(prefix + ".logs", name), "rb").read() == open(str(tmp_path / "oracle_logs" / name), "rb").read()


@pytest.mark.parametrize("engine", ENGINES)
def test_cpp_dockermc_chooses_the_oracles_moves(engine, tmp_path):
    """`class DockerMC` (corintho_ai_amd/cpp/dockermc.h = dockermc.h:13-51) driven from C++ by the loop of
    docker/choose_move.pyx: move, done / drawn, node count, evaluation and legal moves equal the oracle's DockerMC"""
    from tests.test_analyse import _oracle_result, _positions

    libdir, libname = _lib_for(engine)
    exe = str(tmp_path / ("dockermc_driver_" + engine))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cxx", "dockermc_driver.cpp"),
                           "-L" + libdir, "-l" + libname, "-Wl,-rpath," + libdir, "-fopenmp"])
    n, S_, spe = 10, 60, 8
    boards, tp, pc = _positions(n, seed=77)
    seeds = [11 * i + 5 for i in range(n)]
    text = "".join("%d %d %s %s\n" % (seeds[i], tp[i], " ".join(map(str, pc[i])), " ".join(map(str, boards[i]))) for i in range(n))
    out = subprocess.run([exe, str(S_), str(spe)], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    for i in range(n):
        want, _ = _oracle_result(boards[i], tp[i], pc[i], seeds[i], S_, spe, H.hash_net)
        if "pre-result" in want:
            assert out[i] == "pre " + want["pre-result"]
            continue
        move, done, drawn, nodes, eb, m0, m1, m2 = [int(x) for x in out[i].split()]
        mask = m0 | (m1 << 32) | (m2 << 64)
        assert move == want["move"] and bool(done) == want["is_done"] and nodes == want["nodes_searched"]
        assert bool(done and not drawn) == want["has_won"]
        assert [j for j in range(96) if mask >> j & 1] == (want["legal_moves"] if not done else [j for j in range(96) if mask >> j & 1])
        assert np.array([eb], np.uint32).view(np.float32)[0] == np.float32(want["eval_sum"])


def test_reference_main_pyx_compiles_against_this_trainer(tmp_path):
    """main.pyx:17 `cdef extern from "../cpp/src/trainer.cpp"` -> this repository's trainer.cpp; everything
    else of the reference's Cython module unchanged.  Cythonize + compile + link (python/setup.py:15-38 flags)."""
    pyx = os.path.join(REFERENCE, "corintho_ai/python/main.pyx")
    if not os.path.exists(pyx):
        pytest.skip("reference tree not mounted")
    try:
        import Cython  # noqa: F401
    except ImportError:
        pytest.skip("cython not installed")
    from corintho_ai_amd import build

    build.build()
    text = open(pyx).read()
    new_path = os.path.join(ROOT, "corintho_ai_amd", "cpp", "trainer.cpp")
    patched, n = re.subn(r'cdef extern from "\.\./cpp/src/trainer\.cpp"', 'cdef extern from "%s"' % new_path, text)
    assert n == 1
    (tmp_path / "main.pyx").write_text(patched)
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "main.pyx"], cwd=tmp_path)
    inc = sysconfig.get_paths()["include"]
    libdir = os.path.join(ROOT, "corintho_ai_amd")
    so = str(tmp_path / "main_ext.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fopenmp", "-DNDEBUG", "-shared", "-fPIC", "-I" + inc, "-I" + np.get_include(),
                           "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION", "main.cpp", "-o", so, "-L" + libdir, "-lcorintho_hip",
                           "-Wl,-rpath," + libdir], cwd=tmp_path)
    # the extension defines the module init and calls the engine's C ABI through the Trainer members
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", so]).decode()
    for s in ("ca_trainer_create", "ca_trainer_do_iteration", "ca_trainer_write_requests", "ca_trainer_write_samples",
              "ca_trainer_num_requests", "ca_trainer_write_scores"):
        assert s in syms, s
    assert "PyInit_main" in subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
