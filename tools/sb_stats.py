"""Counters of the grouped search (mcts.h co_search_batch) on the emulation build: how many simulations of a step end
the ORDINARY way (and are committed in groups), what ends a group early, how often two simulations of a group meet in
one node below the root.  CPU only:

    python tools/sb_stats.py [games] [sims] [net]     # net: mlp (random init) | trained (a reference checkpoint)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libcorintho_emu_sbstats.so"


def build(sb):
    csrc = os.path.join(ROOT, "corintho_ai_amd", "csrc")
    cmd = ["g++", "-O2", "-std=c++17", "-DCO_EMU", "-DCO_SB_STATS", "-DCO_SB=%d" % sb, "-ffp-contract=off", "-fno-fast-math",
           "-fopenmp", "-fPIC", "-Wno-unknown-pragmas", "-shared", "-o", LIB, "-x", "c++", os.path.join(csrc, "engine.hip"),
           "-x", "c++", os.path.join(ROOT, "tests", "emu", "nn_emu.cpp"), "-lm"]
    subprocess.check_call(cmd)


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    sims = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    net = sys.argv[3] if len(sys.argv) > 3 else "mlp"
    sb = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    build(sb)
    os.environ["CO_EMU_LIB"] = LIB
    import numpy as np

    from corintho_ai_amd import NET_MLP12X100, nets
    from tests import engines as E

    if net == "trained":
        d = np.load(os.path.join(ROOT, "tests", "golden", "trained_last.npz"))
        w = d["weights"] if "weights" in d else d[d.files[0]]
    else:
        w = nets.init_mlp12x100(0)
    t = E.make_trainer("emu", games, "", 12345, sims, 16, 1.0, 0.25, 0, 1, False, stagger=False)
    t.set_net(NET_MLP12X100, w)
    assert t.run()
    L = E.cdll("emu")
    L.co_emu_sb_stats.restype = C.POINTER(C.c_ulonglong)
    s = [int(L.co_emu_sb_stats()[i]) for i in range(32)]
    st = t.stats()
    tot = s[2] + s[8]
    print("games %d, sims/move %d, net %s, CO_SB %d: %d simulations (engine counts %d)" % (games, sims, net, sb, tot, st["searches"]))
    print("  groups %d, asked %d, committed in groups %d (%.1f %%), sequential %d (%.1f %%)" %
          (s[0], s[1], s[2], 100.0 * s[2] / tot, s[8], 100.0 * s[8] / tot))
    print("  group ended by: nothing searchable %d, terminal child %d, wide node %d, too deep %d, terminal leaf %d" %
          (s[3], s[4], s[5], s[6], s[7]))
    print("  levels per group %.2f, scans per committed simulation %.2f, patches below the root per scan %.3f" %
          (s[9] / max(s[0], 1), s[10] / max(s[2], 1), s[11] / max(s[10], 1)))
    print("  levels below the root per group: shared %.2f, wave-wide for one row %.2f / two rows %.2f, row form %.2f (%.2f turns per such level)" %
          (s[20] / max(s[0], 1), s[21] / max(s[0], 1), s[22] / max(s[0], 1), s[23] / max(s[0], 1), s[24] / max(s[23], 1)))
    L.co_emu_sb_ply.restype = C.POINTER(C.c_ulonglong)
    pl = [int(L.co_emu_sb_ply()[i]) for i in range(64)]
    print("  by game progress (plies / 4): groups, committed per group, groups ended by a terminal leaf / by the sequential path, sequential simulations")
    for b in range(8):
        g = pl[8 * b]
        if g:
            print("    plies %2d-%2d: %7d groups, %.2f committed, %4.1f %% terminal, %4.1f %% sequential, %d sequential simulations" %
                  (4 * b, 4 * b + 3, g, pl[8 * b + 2] / g, 100.0 * pl[8 * b + 3] / g, 100.0 * pl[8 * b + 4] / g, pl[8 * b + 6]))
    print("  groups by simulations committed: " + ", ".join("%d: %d" % (k, s[12 + k]) for k in range(sb + 1)))


if __name__ == "__main__":
    main()
