#!/usr/bin/env python3
"""Stress of the shared evaluation cache's emptying protocol under real concurrency: small tables (emptied every few
iterations) with 2 and 3 pools on the GPU, every game of the generation replayed on the oracle bit for bit.
usage (GPU box): python tools/exp/cache_stress.py [games]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from corintho_ai_amd import NET_RESCNN4_H3, nets  # noqa: E402
from tests.engines import make_trainer  # noqa: E402
from tests.test_configs_gpu import _replay_whole_generation  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for pools, bits, seed in ((2, 14, 1), (3, 14, 2), (3, 17, 3), (2, 20, 4), (3, 0, 5)):
    t = make_trainer("hip", G, "", seed, 400, 16, 1.0, 0.25, 0, 1, False, stagger=False, pools=pools, eval_cache=bits if bits else True)
    t.set_net(NET_RESCNN4_H3, nets.init_rescnn4(0))
    assert t.run()
    st = t.stats()
    _replay_whole_generation(t, G, 400, 16, seed)
    print("%d games, %d pools, table 2^%s: %d of %d rows evaluated, %s emptyings -- every game bit-exact on the oracle"
          % (G, pools, bits or "auto", st["nn_rows_evaluated"], st["nn_rows"], st.get("cache_clears", "?")), flush=True)
    t.close()
