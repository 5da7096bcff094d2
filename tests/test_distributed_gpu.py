"""The device half of the multi-GPU path on the ONE GPU a test box has: a process group of one rank
over `nccl` (= RCCL), `ca_trainer_pack_samples_device` into torch-owned HBM, `all_gather_into_tensor` on CUDA tensors
-- what `Trainer::writeSamples` (trainer.cpp:103-113) is to one process, the gather is to N.  `bench.py --gpus 1` is
started as a FRESH child process with CORINTHO_FORCE_DIST=1 (never a re-exec of this one: a process that has touched the
GPU must not be replaced)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_gather_of_one_rank_on_the_device():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, CORINTHO_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--games", "512", "--sims", "100", "--steps", "2",
                        "--warmup", "1", "--net", "mlp12x100h3", "--cpu-games", "0", "--no-variants", "--no-unshared",
                        "--recycle-games", "0", "--check-gather"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # RCCL's banner must not land on stdout
    out = json.loads(lines[0])
    d = out["detail"]
    assert out["n_gpus"] == 1 and d["world_size"] == 1 and d["ranks_seen"] == 1
    c = d["collectives_per_step"]
    assert c["samples_gathered"] == c["samples_of_all_shards"] > 0 and c["unfinished_games"] == 0
    k = c["gather_check"]
    assert k["backend"] == "nccl" and k["payload_on"] == "cuda"
    assert k["rows_gathered"] == k["rows_own"] == k["num_samples"] > 512  # more than a ply per game
    assert k["bytes_equal_to_export_samples"] and k["ranks_with_a_difference"] == 0
    assert k["payload_bytes"] == k["rows_gathered"] * 167 * 4
    assert d["rank0_device_ms_per_step"]["sample_gather_and_score_allreduce"] > 0
    print("RCCL one-rank gather: %d rows, %.2f ms gather + copy back, %.2f ms per step in the timed region"
          % (k["rows_gathered"], k["gather_and_copy_back_ms"], d["rank0_device_ms_per_step"]["sample_gather_and_score_allreduce"]))
