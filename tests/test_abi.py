"""CPU checks of the drop-in boundary: the HIP library loads and exports every
symbol include/corintho_hip.h declares (no compute calls here), the C++ `Trainer`
wrapper compiles and links against it, and a missing library fails loudly."""
import ctypes as C
import os
import re
import subprocess

import pytest

from corintho_ai_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include/corintho_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ca_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from corintho_ai_amd import build

    build.build()
    L = C.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "libcorintho_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms, "corintho_ai_amd/_lib.py EXPORTS out of sync with the header"


def test_no_gpu_is_an_error_not_a_fallback():
    """without a gfx950 device the product refuses to run (no CPU path)"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from corintho_ai_amd import Trainer

    with pytest.raises(_lib.EngineError):
        Trainer(4, "", 1, 8, 4)


def test_cpp_trainer_wrapper_compiles_and_links(tmp_path):
    """corintho_ai_amd/cpp/trainer.{h,cpp}: the reference's class surface over the C ABI"""
    from corintho_ai_amd import build

    build.build()
    src = tmp_path / "use.cpp"
    src.write_text(
        '#include "%s/corintho_ai_amd/cpp/trainer.cpp"\n'
        "int main(int argc, char**) {\n"
        "  if (argc > 100) {  // type-check the reference call pattern (main.pyx:299-310, 142-168)\n"
        '    Trainer t(8, "logs", 1, 50, 16, 1.0f, 0.25f, 0, 1, false);\n'
        "    float e[128], p[128 * 96], g[128 * 70];\n"
        "    while (!t.doIteration(e, p, -1)) { int n = t.num_requests(-1); (void)n; t.writeRequests(g, -1); }\n"
        "    t.num_samples(); t.score(); t.avg_mate_length(); t.writeSamples(g, e, p); t.writeScores(\"x\");\n"
        "  }\n"
        "  Trainer d;  // default ctor must exist (trainer.h:19-21)\n"
        "  return 0;\n"
        "}\n" % ROOT
    )
    exe = tmp_path / "use"
    subprocess.check_call(["g++", "-std=c++17", "-o", str(exe), str(src), "-L" + os.path.dirname(_lib.LIB_PATH),
                           "-lcorintho_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    subprocess.check_call([str(exe)])


def test_cpp_tourney_wrapper_compiles_and_links(tmp_path):
    """corintho_ai_amd/cpp/tourney.{h,cpp}: the reference's Tourney surface over the C ABI"""
    from corintho_ai_amd import build

    build.build()
    src = tmp_path / "use.cpp"
    src.write_text(
        '#include "%s/corintho_ai_amd/cpp/tourney.cpp"\n'
        "int main(int argc, char**) {\n"
        "  if (argc > 100) {  // type-check the reference call pattern (rating/tourney.pyx:63-160)\n"
        '    Tourney t(4, "logs");\n'
        "    t.addPlayer(0, 0, 1600, 16, 1.0f, 0.25f, false); t.addPlayer(1, -1, 0, 0, 1.0f, 0.25f, true);\n"
        "    t.addMatch(0, 1, false);\n"
        "    float e[64], p[64 * 96], g[64 * 70];\n"
        "    while (!t.all_done()) { if (t.num_requests(0) > 0) t.writeRequests(g, 0); t.doIteration(e, p, 0); }\n"
        '    t.writeScores("scores.txt");\n'
        "  }\n"
        "  return 0;\n"
        "}\n" % ROOT
    )
    exe = tmp_path / "use"
    subprocess.check_call(["g++", "-std=c++17", "-o", str(exe), str(src), "-L" + os.path.dirname(_lib.LIB_PATH),
                           "-lcorintho_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH)])
    subprocess.check_call([str(exe)])
