"""Experiment helper: libcorintho_hip built with -DCO_WINOGRAD (kind 7 = the Winograd variant of rescnn4x6,
tools/exp/nn_rescnn_wino.inc) into build_ab/, never the product library."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import _lib, build  # noqa: E402

KIND_W6 = 7


def load(extra=()):
    out = os.path.join(ROOT, "build_ab", "libcorintho_hip_wino.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([build.hipcc()] + build.FLAGS + ["-DCO_WINOGRAD"] + list(extra) + ["-o", out] +
                          [os.path.join(build.CSRC, s) for s in build.SOURCES])
    return _lib.declare(C.CDLL(out))
