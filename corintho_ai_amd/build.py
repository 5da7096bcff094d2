"""Build recipe of libcorintho_hip.so: hipcc, gfx950 only, in-tree.

    python -m corintho_ai_amd.build        # or __graft_entry__.build()

-ffp-contract=off and correctly rounded fp32 divide/sqrt are part of the
correctness contract of the search kernels (csrc/mcts.h), not tuning flags.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcorintho_hip.so")
SOURCES = ["engine.hip", "nn_mlp.hip", "nn_mlp_split.hip", "nn_rescnn.hip"]
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function",
    "-Wno-unused-variable", "-Wno-unknown-pragmas",
]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "corintho_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra) + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
