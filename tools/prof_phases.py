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
cmd = [build.hipcc()] + build.FLAGS + ["-DCO_PROF", "-o", out] + [os.path.join(build.CSRC, s) for s in build.SOURCES]
subprocess.check_call(cmd)
L = _lib.declare(C.CDLL(out))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
t = Trainer(G, "", 12345, S, 16, 1.0, 0.25, 0, 1, False, stagger=False, _cdll=L)
t.set_net(9, nets.init_mlp12x100(0))
t.run()
st = t.stats()
p = (C.c_ulonglong * 36)()
L.ca_trainer_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
_lib.check(L, L.ca_trainer_prof(t._t, p))
rec, srch, choose, expand, steps, nsearch, nrecv = [int(x) for x in p[:7]]
print("stats", {k: st[k] for k in ("searches", "evals", "iterations", "mcts_ms", "nn_ms", "pack_ms")})
print("per receive: %.0f cycles   (%d receives)" % (rec / max(nrecv, 1), nrecv))
print("per search : %.0f cycles   (%d searches), of which expand %.0f" % (srch / max(nsearch, 1), nsearch, expand / max(nsearch, 1)))
print("choose/hand-over per step: %.0f cycles over %d wave-steps" % (choose / max(steps, 1), steps))
wave_total = int(p[7])
print("whole wave-step: %.0f cycles avg; clock of game 1's steps: %.0f MHz" % (wave_total / max(steps, 1), 100.0 * int(p[16]) / max(int(p[17]), 1)))
print("launch avg %.1f us -> %.0f cycles at that clock" % (1e3 * st["mcts_ms"] / st["iterations"], 1e3 * st["mcts_ms"] / st["iterations"] * int(p[16]) / max(int(p[17]), 1) * 100))
print("max wave-step %d cycles; histogram of wave-step cycles (50k buckets): %s" % (int(p[18]), [int(x) for x in p[20:36]]))
print('phases of a simulation (cycles per search): PUCT scans %.0f, slot stores %.0f, expansion %.0f, descent block fetch (waited for) %.0f, '
      'terminal leaves %.0f, request %.0f' % tuple(int(p[i]) / max(nsearch, 1) for i in range(8, 14)))
tot = rec + srch + choose
print("share: receive %.1f%%  search %.1f%%  choose %.1f%%;  stamped cycles per wave-step %.0f" %
      (100 * rec / tot, 100 * srch / tot, 100 * choose / tot, tot / max(steps, 1)))
