#!/usr/bin/env python3
"""Self-play throughput of the MI355X engine (BASELINE.json metric:
"self-play games/sec at 400 sims/move").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one whole self-play generation of `--games` games per GPU in fused mode
(search + network on the device), from the first iteration to the last game
finishing, plus -- for N > 1 -- the RCCL all-gather of the un-augmented samples.
Weak scaling: every rank owns `--games` games, a contiguous shard of a generation
of N x games (seeds and colours follow the global game index, trainer.cpp:243-255).
The timed region is bracketed by barrier + device synchronisation; the time is the
max over ranks; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1] / SURVEY 8d): 4096 games per GPU, 400
simulations per move, 16 searches per evaluation, c_puct 1.0, epsilon 0.25,
random-init weights (seed 0), synthetic = self-generated positions.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters (dense fp32 matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # ibid. (dense bf16 matrix)
HBM_PEAK_GBS = 8000.0          # ibid.
BYTES_PER_SIM = 3200.0         # algorithmic bytes per simulation at 400 sims/move


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--spe", type=int, default=16)
    ap.add_argument("--c-puct", type=float, default=1.0)
    ap.add_argument("--epsilon", type=float, default=0.25)
    ap.add_argument("--net", default="rescnn4x3", choices=["mlp12x100", "mlp12x100x3", "rescnn4", "rescnn4x3"],
                    help="rescnn4x3 (default) = the 4-block residual CNN BASELINE.json configs[1] names, convolutions at "
                         "bf16x3 split precision (within 2e-5 of fp32); rescnn4 = the same network on fp32 MFMA; "
                         "mlp12x100 = the reference's own net")
    ap.add_argument("--no-mlp-extra", action="store_true", help="skip the extra generations (other networks) reported under detail.variants")
    ap.add_argument("--stagger", action="store_true", help="keep the reference's staggered start")
    ap.add_argument("--tflite", default="", help="with --net mlp12x100: weights imported from a reference TFLite checkpoint")
    ap.add_argument("--arena-units", type=int, default=0)
    ap.add_argument("--no-unshared", action="store_true", help="skip the extra single-pool generation behind roofline.unshared (profiling runs)")
    ap.add_argument("--pools", type=int, default=0, help="independent game pools on separate streams per GPU (0 = engine default)")
    ap.add_argument("--cpu-games", type=int, default=64, help="games of the bounded CPU-baseline sample (0 = skip)")
    return ap.parse_args()


def measured_traffic(kernel, args, npools):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_f_pmc.json; PMC counters cannot be collected from inside an un-profiled run).
    Only valid for the workload those passes were taken on (the default one); otherwise null."""
    if (args.games, args.sims, args.spe, args.net, npools) != (4096, 400, 16, "rescnn4x3", 2):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_f_pmc.json")) as f:
            k = json.load(f)["kernels"]
            t = k[kernel]["traffic_bytes_per_launch"]
            if kernel + "_small" in k:  # the network launch queues both instances of the kernel; one of them works
                t += k[kernel + "_small"]["traffic_bytes_per_launch"]
            return t
    except Exception:
        return None


def host_cores():
    """cores this process may really use: affinity, cgroup quota, and the GPU box's
    per-GPU CPU share (16) -- os.cpu_count() reports the whole host"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("CORINTHO_CPU_THREADS", "16"))))


def cpu_baseline(args, weights, net_name="mlp12x100"):
    """The oracle (CPU restatement of the reference's OpenMP path) on this box's host
    cores, with the same network evaluated on the CPU between iterations as the
    reference's Keras loop does (main.pyx:70-83).  Bounded sample; rank 0, N = 1 only."""
    from corintho_ai_amd import nets
    from oracle import oracle as O
    from tests import harness as H

    cores = host_cores()
    G = args.cpu_games
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=cores)  # numpy's BLAS on the same cores
    except Exception:
        pass
    t = O.Trainer(G, seed=12345, max_searches=args.sims, searches_per_eval=args.spe, c_puct=args.c_puct,
                  epsilon=args.epsilon, num_threads=cores)
    t.set_stagger(False)  # as the GPU run
    nn_time = [0.0]

    if net_name == "rescnn4":
        import torch

        torch.set_num_threads(cores)
        fwd = nets.rescnn4_forward_ref
    else:
        fwd = nets.mlp12x100_forward_np

    def net(states):
        t0 = time.perf_counter()
        out = fwd(weights, states)
        nn_time[0] += time.perf_counter() - t0
        return out

    t0 = time.perf_counter()
    H.play_generation(t, G, args.spe, net)
    dt = time.perf_counter() - t0
    # the search alone on a larger sample (enough games to keep every thread busy), with a
    # zero-cost stand-in network: the upper bound of the CPU path whatever the inference costs
    G2 = 16 * G
    t2 = O.Trainer(G2, seed=12345, max_searches=args.sims, searches_per_eval=args.spe, c_puct=args.c_puct,
                   epsilon=args.epsilon, num_threads=cores)
    t2.set_stagger(False)
    nn2 = [0.0]

    def fake(states):
        t1 = time.perf_counter()
        out = H.uniform_net(states)
        nn2[0] += time.perf_counter() - t1
        return out

    t0 = time.perf_counter()
    H.play_generation(t2, G2, args.spe, fake)
    dt2 = time.perf_counter() - t0 - nn2[0]
    return {
        "value": G / dt,
        "unit": "games/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d games x %d sims/move, spe %d, oracle/ (OpenMP, %d threads) + fp32 %s on the host (%s); "
                  "%.1f s total, %.1f s of it network" % (G, args.sims, args.spe, cores, net_name,
                                                          "torch CPU" if net_name == "rescnn4" else "numpy", dt, nn_time[0]),
        "mcts_only_games_per_s": G2 / max(dt2, 1e-9),
        "mcts_only_sample": "%d games, uniform stand-in network (its cost excluded), %.1f s" % (G2, dt2),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    torch = None
    use_dist = world > 1 or os.environ.get("CORINTHO_FORCE_DIST") == "1"  # the latter: 1-rank rehearsal of the RCCL path
    if use_dist:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from corintho_ai_amd import NET_MLP12X100, NET_MLP12X100_X3, NET_RESCNN4, NET_RESCNN4_X3, Trainer, nets

    G = args.games
    if args.net in ("mlp12x100", "mlp12x100x3"):
        if args.tflite:
            from corintho_ai_amd.tflite_import import mlp12x100_from_tflite

            weights = mlp12x100_from_tflite(args.tflite)
        else:
            weights = nets.init_mlp12x100(0)
        kind = NET_MLP12X100_X3 if args.net == "mlp12x100x3" else NET_MLP12X100
        flop_per_row = 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96)
    else:
        weights = nets.init_rescnn4(0)
        kind = NET_RESCNN4_X3 if args.net == "rescnn4x3" else NET_RESCNN4
        flop_per_row = nets.rescnn4_flop_per_row()

    tr = Trainer(G, "", 12345, args.sims, args.spe, args.c_puct, args.epsilon, 0, 1, False, device=local_rank,
                 stagger=args.stagger, arena_units=args.arena_units, game_base=rank * G, total_games=world * G,
                 pools=args.pools)
    tr.set_net(kind, weights)

    gatherer = None
    if use_dist:
        from corintho_ai_amd.dist import SampleGather

        gatherer = SampleGather(tr, G, on_device=True)

    totals = {"searches": 0, "evals": 0, "plies": 0, "iterations": 0, "mcts_ms": 0.0, "nn_ms": 0.0, "pack_ms": 0.0,
              "nn_rows": 0, "nn_launches": 0, "mcts_launches": 0, "timed_launches": 0, "nn_timed_rows": 0,
              "mcts_timed_ms": 0.0, "nn_timed_ms": 0.0, "gather_ms": 0.0, "samples": 0, "peak_arena_units": 0}

    def one_step(step_index, timed):
        tr.reset(12345 + step_index)
        done = tr.run()
        if not done:
            raise RuntimeError("generation did not finish")
        if use_dist:
            t0 = time.perf_counter()
            gatherer.gather()  # one RCCL all-gather of the un-augmented samples per generation
            if timed:
                totals["gather_ms"] += (time.perf_counter() - t0) * 1e3
        if timed:
            st = tr.stats()
            for k in ("searches", "evals", "plies", "iterations", "mcts_ms", "nn_ms", "pack_ms", "nn_rows", "nn_launches",
                      "mcts_launches", "timed_launches", "nn_timed_rows", "mcts_timed_ms", "nn_timed_ms"):
                totals[k] += st[k]
            totals["pools"] = st["pools"]
            totals["samples"] += tr.num_samples()
            totals["peak_arena_units"] = max(totals["peak_arena_units"], st["peak_arena_units"])

    def barrier():
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        one_step(1000 + w, False)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(s, True)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        agg = torch.tensor([totals["searches"], totals["evals"], totals["nn_rows"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        job_searches, job_evals, job_rows = [float(x) for x in agg.tolist()]
    else:
        job_searches, job_evals, job_rows = float(totals["searches"]), float(totals["evals"]), float(totals["nn_rows"])

    if rank == 0:
        games_total = world * G * args.steps
        value = games_total / dt
        nn_s = totals["nn_ms"] * 1e-3
        mcts_s = totals["mcts_ms"] * 1e-3
        npools = int(totals.get("pools", 1))
        # dominant kernel = the family with more device time on rank 0
        # The fused loop times one iteration per pool and window of 8 with HIP events on the pool's
        # stream (timing every launch costs 1-4 % of the wall time); the network kernel's rate is
        # computed on exactly those launches: their batch rows and their durations.
        tl = totals["timed_launches"]
        if nn_s >= mcts_s:
            if tl > 0:
                achieved = totals["nn_timed_rows"] * flop_per_row / max(totals["nn_timed_ms"] * 1e-3, 1e-12) / 1e12
            else:
                achieved = totals["nn_rows"] * flop_per_row / max(nn_s, 1e-12) / 1e12
            peak = BF16_MFMA_PEAK_TFLOPS if args.net.endswith("x3") else FP32_MFMA_PEAK_TFLOPS
            kname = {"mlp12x100": "co_k_mlp_forward", "mlp12x100x3": "co_k_mlp_forward_x3", "rescnn4": "co_k_rescnn_forward",
                     "rescnn4x3": "co_k_rescnn_forward_x3"}[args.net]
            roofline = {"kernel": kname,
                        "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "traffic": measured_traffic(kname, args, npools),
                        "issued_frac": (3.0 if args.net.endswith("x3") else 1.0) * achieved / peak,
                        "algorithmic": "%.1f KFLOP/row x %d rows in %d timed launches of %d" %
                                       (flop_per_row / 1e3, totals["nn_timed_rows"] if tl else totals["nn_rows"],
                                        tl if tl else totals["nn_launches"], totals["nn_launches"]),
                        "avg_launch_ms": (totals["nn_timed_ms"] / tl) if tl else totals["nn_ms"] / max(totals["nn_launches"], 1)}
        else:
            achieved = totals["searches"] * BYTES_PER_SIM / max(mcts_s, 1e-12) / 1e9
            roofline = {"kernel": "co_k_mcts_step", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": measured_traffic("co_k_mcts_step", args, npools),
                        "algorithmic": "%.0f B/simulation x %d simulations" % (BYTES_PER_SIM, totals["searches"]),
                        "avg_launch_ms": (totals["mcts_timed_ms"] / tl) if tl else totals["mcts_ms"] / max(totals["mcts_launches"], 1)}
        roofline["streams"] = npools
        if npools > 1 and not args.no_unshared and world == 1:
            # The timed region runs the games as `npools` pools on separate streams, so the durations
            # above are those of kernels SHARING the GPU with the other pool's kernels (their sum
            # exceeds the wall time).  One more generation with a single pool gives the same kernel's
            # rate when it has the GPU to itself.
            t1 = Trainer(G, "", 12345, args.sims, args.spe, args.c_puct, args.epsilon, 0, 1, False, device=local_rank,
                         stagger=args.stagger, arena_units=args.arena_units, game_base=rank * G,
                         total_games=world * G, pools=1)
            t1.set_net(kind, weights)
            t1.reset(1000)
            t1.run()
            t1.reset(0)
            tu = time.perf_counter()
            t1.run()
            du = time.perf_counter() - tu
            su = t1.stats()
            if roofline["kernel"] == "co_k_mcts_step":
                a1 = su["searches"] * BYTES_PER_SIM / max(su["mcts_ms"] * 1e-3, 1e-12) / 1e9
                l1 = su["mcts_ms"] / max(su["mcts_launches"], 1)
            else:
                if su["timed_launches"] > 0:
                    a1 = su["nn_timed_rows"] * flop_per_row / max(su["nn_timed_ms"] * 1e-3, 1e-12) / 1e12
                    l1 = su["nn_timed_ms"] / su["timed_launches"]
                else:
                    a1 = su["nn_rows"] * flop_per_row / max(su["nn_ms"] * 1e-3, 1e-12) / 1e12
                    l1 = su["nn_ms"] / max(su["nn_launches"], 1)
            roofline["unshared"] = {"achieved": a1, "frac": a1 / roofline["peak"], "avg_launch_ms": l1,
                                    "games_per_s_single_pool": G / du,
                                    "note": "same kernel, same workload, one pool on one stream (nothing else on the GPU)"}
            del t1
        out = {
            "metric": "self-play games/sec at %d sims/move" % args.sims,
            "value": value,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16x3" if args.net.endswith("x3") else "f32",
            "data": "synthetic",
            "config": {
                "workload": "%d parallel self-play games per GPU, %d sims/move, %d searches/eval, %s (random init, seed 0), "
                            "fused on-device search + inference, %d pool(s) per GPU, %s"
                            % (G, args.sims, args.spe, args.net, npools, "staggered start" if args.stagger else "no stagger"),
                "games_per_gpu": G, "sims_per_move": args.sims, "searches_per_eval": args.spe, "net": args.net,
                "c_puct": args.c_puct, "epsilon": args.epsilon, "parallelism": "games sharded x%d" % world,
            },
            "roofline": roofline,
            "detail": {
                "sims_per_s": job_searches / dt,
                "leaf_evals_per_s": job_evals / dt,
                "plies_per_game": totals["plies"] / max(G * args.steps, 1),
                "evals_per_game": totals["evals"] / max(G * args.steps, 1),
                "iterations_per_step": totals["iterations"] / max(args.steps, 1),
                "rank0_device_ms_per_step": {"mcts": totals["mcts_ms"] / args.steps, "network": totals["nn_ms"] / args.steps,
                                             "pack": totals["pack_ms"] / args.steps,
                                             "sample_gather": totals["gather_ms"] / args.steps},
                "mcts_GBps_algorithmic": totals["searches"] * BYTES_PER_SIM / max(mcts_s, 1e-12) / 1e9,
                "network_TFLOPs_algorithmic": totals["nn_rows"] * flop_per_row / max(nn_s, 1e-12) / 1e12,
                "peak_arena_units_per_tree": totals["peak_arena_units"],
            },
        }
        if world == 1 and not args.no_mlp_extra:
            # the other networks on the same pool, one timed generation each, for comparison
            variants = {}
            for name, vkind, vw, vflop, vpeak in (
                ("rescnn4_fp32", NET_RESCNN4, None, nets.rescnn4_flop_per_row(), FP32_MFMA_PEAK_TFLOPS),
                ("rescnn4_bf16x3", NET_RESCNN4_X3, None, nets.rescnn4_flop_per_row(), BF16_MFMA_PEAK_TFLOPS),
                ("mlp12x100_fp32", NET_MLP12X100, "mlp", 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96),
                 FP32_MFMA_PEAK_TFLOPS),
                ("mlp12x100_bf16x3", NET_MLP12X100_X3, "mlp", 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96),
                 BF16_MFMA_PEAK_TFLOPS),
            ):
                if vkind == kind:
                    continue
                tr.set_net(vkind, nets.init_mlp12x100(0) if vw == "mlp" else nets.init_rescnn4(0))
                tr.reset(777)
                tr.run()  # warm
                tr.reset(778)
                t1 = time.perf_counter()
                tr.run()
                d1 = time.perf_counter() - t1
                st = tr.stats()
                tf = st["nn_rows"] * vflop / max(st["nn_ms"] * 1e-3, 1e-12) / 1e12
                variants[name] = {"games_per_s": G / d1, "ms_per_step": d1 * 1e3, "network_TFLOPs_algorithmic": tf,
                                  "network_frac_of_mfma_peak": tf / vpeak,
                                  "device_ms": {"mcts": st["mcts_ms"], "network": st["nn_ms"], "pack": st["pack_ms"]}}
            out["detail"]["variants"] = variants
        if world == 1 and args.cpu_games > 0:
            out["cpu_baseline"] = cpu_baseline(args, weights, "mlp12x100" if args.net.startswith("mlp12x100") else "rescnn4")
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
