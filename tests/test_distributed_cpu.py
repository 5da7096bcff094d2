"""Tests of the sharded path on CPU at world sizes 2 and 8 (the size BASELINE configs[2] asks for): one process per rank
(gloo, 127.0.0.1), each running its shard of one generation on the emulation build of the engine, then the
per-generation sample all-gather; the gathered, x8-expanded samples must equal what ONE trainer over all games writes
(Trainer::writeSamples order, trainer.cpp:103-113; seeds and colours on the global index, :243-255)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
from corintho_ai_amd import Trainer, expand_samples, nets
from corintho_ai_amd.dist import SampleGather, shard
from tests.emu import emulib
from tests import harness as H
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
L = emulib.load()
G, S, spe = %(games)d, 24, 8
base, total = shard(rank, world, G)
t = Trainer(G, "", 4242, S, spe, 1.0, 0.25, 0, 1, False, stagger=False, game_base=base, total_games=total, _cdll=L)
t.set_net(1, nets.init_mlp12x100(0, bn_noise=True))
assert t.run()
g = SampleGather(t, G, on_device=False)
sp, oc = g.rows()
score, unfinished = g.score_and_unfinished(True)   # C2: one all-reduce of two scalars
assert unfinished == 0
if rank == 0:
    gs, ev, pr = expand_samples(sp, oc, _cdll=L)
    np.savez(%(out)r, gs=gs, ev=ev, pr=pr, score=score, moved=g.bytes_moved, rows=sp.shape[0])
dist.destroy_process_group()
"""


def _free_port():
    """a port nobody holds right now (two test runs on one host must not meet on a fixed one: ADVICE round 5)"""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


@pytest.mark.parametrize("world,games", [(2, 6), (8, 3)], ids=["2_ranks", "8_ranks"])
def test_sharded_generation_matches_single_trainer(tmp_path, world, games):
    port = _free_port()
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": out, "games": games})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, WORLD_SIZE=str(world), OMP_NUM_THREADS="1" if world > 2 else "2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    # the same generation on one trainer
    from corintho_ai_amd import Trainer, nets
    from tests import harness as H
    from tests.emu import emulib

    t = Trainer(world * games, "", 4242, 24, 8, 1.0, 0.25, 0, 1, False, stagger=False, _cdll=emulib.load())
    t.set_net(1, nets.init_mlp12x100(0, bn_noise=True))
    assert t.run()
    gs, ev, pr = H.get_samples(t)
    assert got["gs"].tobytes() == gs.tobytes()
    assert got["ev"].tobytes() == ev.tobytes()
    assert got["pr"].tobytes() == pr.tobytes()
    assert abs(float(got["score"]) - t.score()) < 1e-6
    # the payload all-gather is sized by the largest shard, not by the 44-ply upper bound
    assert int(got["moved"]) < world * (int(got["rows"]) / world) * 167 * 4 * (1.5 if world == 2 else 2.5)


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


TINY = ["--engine", "emu", "--games", "6", "--sims", "24", "--spe", "8", "--steps", "1", "--warmup", "0", "--net", "mlp12x100",
        "--cpu-games", "0", "--no-variants", "--no-unshared", "--recycle-games", "0"]


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls it): two rank processes, one
    JSON line, n_gpus == 2, both ranks seen, the gathered sample count = the sum of the shards (emulation build
    over gloo: the rehearsal of the RCCL path on the GPU-less machine)"""
    import json

    r = _bench(["--gpus", "2", "--check-gather"] + TINY)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    d = out["detail"]
    assert d["world_size"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_games_per_s"]) == 2
    c = d["collectives_per_step"]
    assert c["samples_gathered"] == c["samples_of_all_shards"] > 0 and c["unfinished_games"] == 0
    # every rank found its own export_samples() bytes at its place in what the gather returned
    k = c["gather_check"]
    assert k["bytes_equal_to_export_samples"] and k["ranks_with_a_difference"] == 0 and k["rows_gathered"] > k["rows_own"] > 0
    assert abs(out["value"] - 12 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    # one rank: unchanged single-process path
    r1 = _bench(["--gpus", "1"] + TINY)
    assert r1.returncode == 0, r1.stderr[-2000:]
    o1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert o1["n_gpus"] == 1 and o1["detail"]["ranks_seen"] == 1 and "collectives_per_step" not in o1["detail"]


def test_bench_gpus_8_rehearsal():
    """the eight ranks of BASELINE configs[2], tiny, on the emulation build over gloo: every rank seen, the gathered
    samples = the sum of the eight shards, value = all games over the slowest rank's time"""
    import json

    tiny8 = [a if a != "6" else "2" for a in TINY]  # 2 games per rank
    r = _bench(["--gpus", "8"] + tiny8, env_extra={"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak"
    d = out["detail"]
    assert d["world_size"] == 8 and d["ranks_seen"] == 8 and len(d["per_rank_games_per_s"]) == 8
    c = d["collectives_per_step"]
    assert c["samples_gathered"] == c["samples_of_all_shards"] > 0 and c["unfinished_games"] == 0
    assert abs(out["value"] - 16 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]


def test_bench_ends_every_rank_when_one_dies_during_set_up():
    """a rank other than 0 that dies before the communicator is up: the launcher stops the survivors (rank 0 would
    otherwise wait in the rendezvous until its time-out) and reports the rank and its exit code"""
    import time

    t0 = time.time()
    r = _bench(["--gpus", "2"] + TINY, env_extra={"CORINTHO_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 90
    assert "rank 1 exited with code 3" in r.stderr, r.stderr[-2000:]
    assert not r.stdout.strip()


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _bench(["--gpus", "4"] + TINY, {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
    r = _bench(["--gpus", "2"] + TINY, {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
