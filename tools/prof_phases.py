#!/usr/bin/env python3
"""Diagnostic: build the engine with -DCO_PROF into a separate library and print where
K3's wave cycles go (in-kernel s_memtime stamps).  Never used by the product."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import Trainer, _lib, build, nets  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libcorintho_hip_prof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
csrc = os.environ.get("PROF_CSRC", build.CSRC)  # another tree of the kernel source (an earlier round's, for comparison)
cmd = [build.hipcc()] + build.FLAGS + ["-DCO_PROF"] + extra + ["-o", out] + [os.path.join(csrc, s) for s in build.SOURCES]
subprocess.check_call(cmd)
L = _lib.declare(C.CDLL(out))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
pools = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = Trainer(G, "", 12345, S, 16, 1.0, 0.25, 0, 1, False, stagger=False, pools=pools, _cdll=L)
import numpy as np  # noqa: E402
wts = nets.init_mlp12x100(0)
if os.environ.get("AB_TRAINED"):  # the reference's last checkpoint instead of random init: narrow, deep trees
    wts = np.load(os.path.join(ROOT, "tests", "golden", "trained_last.npz"))["weights"]
t.set_net(9, wts)
t.run()
st = t.stats()
NP = 32
p = (C.c_ulonglong * (2 * NP + 24))()
L.ca_trainer_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
_lib.check(L, L.ca_trainer_prof(t._t, p))
v = [int(x) for x in p]
steps, nsearch, nrecv, levels, nexp = max(v[4], 1), max(v[5], 1), max(v[6], 1), max(v[23], 1), max(v[24], 1)
clk_mhz = 100.0 * v[NP + 0] / max(v[NP + 1], 1)
print("stats", {k: st[k] for k in ("searches", "evals", "iterations", "mcts_ms", "nn_ms", "pack_ms", "pools")})
print("%d wave-steps, %.1f simulations, %.1f evaluations received, %.2f levels scanned per simulation, %.3f expansions per simulation"
      % (steps, nsearch / steps, nrecv / steps, levels / nsearch, nexp / nsearch))
print("whole wave-step: %.0f cycles avg (max %d); clock of game 1's steps: %.0f MHz; launch avg %.1f us = %.0f cycles"
      % (v[7] / steps, v[NP + 2], clk_mhz, 1e3 * st["mcts_ms"] / st["iterations"], 1e3 * st["mcts_ms"] / st["iterations"] * clk_mhz))
print("histogram of wave-step cycles (50k buckets):", v[NP + 4:NP + 20])
names = {22: "wave set-up", 0: "backup: indices", 1: "backup: slot fetch", 3: "backup: sums + stores", 20: "loop control + root load",
         19: "simulation start (root copy, path)", 8: "PUCT scan", 9: "virtual-loss store + path", 11: "descent: block fetch (waited for)",
         16: "expansion: doMove", 14: "expansion: legal moves", 15: "expansion: node stores", 10: "expansion: slot + path update",
         28: "grouped search: root scans", 13: "priors of the pending leaves (rows)", 25: "tail: stores of the step waited for", 26: "tail: batch-row atomic", 27: "tail: noise capture (copy)", 30: "tail: noise capture (before a twist)", 31: "tail: twist", 29: "tail: request rows read back",
         12: "terminal leaf", 17: "request: state row", 18: "request: pending-leaf records", 2: "move choice / hand-over", 21: "step tail"}
tot = sum(v[i] for i in names)
print("stamped %.0f cycles per wave-step (%.1f %% of the whole step); per SIMULATION:" % (tot / steps, 100.0 * tot / max(v[7], 1)))
for i, nm in names.items():
    print("  %-40s %8.0f cycles  %5.1f %%" % (nm, v[i] / nsearch, 100.0 * v[i] / tot))

sv = v[NP + 24:NP + 24 + NP]
ssteps = max(sv[4], 1)
if sv[4]:
    stot = sum(sv[i] for i in names)
    print("SLOW wave-steps (> 280 k cycles): %d (%.2f %% of all), %.0f cycles avg, %.1f simulations, %.1f evaluations received; per wave-STEP:"
          % (sv[4], 100.0 * sv[4] / steps, sv[7] / ssteps, sv[5] / ssteps, sv[6] / ssteps))
    for i, nm in names.items():
        print("  %-40s %8.0f cycles  %5.1f %%   (all steps: %8.0f)" % (nm, sv[i] / ssteps, 100.0 * sv[i] / stot, v[i] / steps))
