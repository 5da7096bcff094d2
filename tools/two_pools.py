#!/usr/bin/env python3
"""Experiment: the 4096-game generation as P independent pools (game shards) driven from P host
threads on P streams of ONE GPU, so that one pool's search kernel overlaps another's network
kernel and launch tails get filled.  usage: two_pools.py [pools] [games_total] [net]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_MLP12X100, NET_RESCNN4, NET_RESCNN4_X3, Trainer, nets  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
net = sys.argv[3] if len(sys.argv) > 3 else "rescnn4x3"
kind = {"mlp12x100": NET_MLP12X100, "rescnn4": NET_RESCNN4, "rescnn4x3": NET_RESCNN4_X3}[net]
w = nets.init_mlp12x100(0) if net == "mlp12x100" else nets.init_rescnn4(0)
g = G // P
pools = []
for p in range(P):
    t = Trainer(g, "", 12345, 400, 16, 1.0, 0.25, 0, 1, False, stagger=False, game_base=p * g, total_games=G)
    t.set_net(kind, w)
    pools.append(t)


def generation(seed):
    for t in pools:
        t.reset(seed)
    th = [threading.Thread(target=t.run) for t in pools]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    return time.perf_counter() - t0


generation(1)
best = min(generation(2 + i) for i in range(3))
print("pools %d x %d games, %s: %.1f ms per generation -> %.0f games/s" % (P, g, net, best * 1e3, G / best))
