#!/usr/bin/env python3
"""Self-play throughput of the MI355X engine (BASELINE.json metric:
"self-play games/sec at 400 sims/move").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one whole self-play generation of `--games` games per GPU in fused mode
(search + network on the device), from the first iteration to the last game
finishing, plus -- for N > 1 -- the RCCL all-gather of the un-augmented samples and the
two-scalar score/done all-reduce.
Weak scaling: every rank owns `--games` games, a contiguous shard of a generation
of N x games (seeds and colours follow the global game index, trainer.cpp:243-255).
The timed region is bracketed by barrier + device synchronisation; the time is the
max over ranks; rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1] / SURVEY 8d): 4096 games per GPU, 400
simulations per move, 16 searches per evaluation, c_puct 1.0, epsilon 0.25,
random-init weights (seed 0), synthetic = self-generated positions.

Arithmetic.  The reference evaluates its network in float32 (Keras / TFLite, main.pyx:70-83);
BASELINE.json asks for policy / value outputs within 1e-4 of float32.  Float32 inputs, weights,
accumulation and epilogues in every kind; what differs is how a matrix product reaches the matrix
pipe (csrc/nn_rescnn.hip, nn_mlp_split.hip):
  * `rescnn4h3` (default): both operands as TWO fp16 terms (22 significand bits), three MFMA products;
    error against float64 within three times the fp32-MFMA kernel's own on every weight set, incl. five
    of the reference's trained checkpoints (measured 0.5-2.3 x; worst case 4e-5 on out-of-distribution
    inputs, 7e-6 on positions met in play);
  * `rescnn4x6`: both operands as THREE bf16 terms whose sum IS the float32 value, six products
    (float32-equivalent; the default of round 2), at twice the matrix work;
  * `rescnn4`: fp32 MFMA;  `*x3`: two bf16 terms, 16 bits -- narrower than float32, variants only.
tests/test_net_precision.py and tests/test_trained_golden.py assert these bounds on the GPU; every
kind is reported under `detail.variants` with its own roofline object.
"""
import argparse
import json
import os
import sys
import time

# the CPU-baseline leg alternates the oracle's OpenMP region with the host network's thread pool on the
# same cores: idle OpenMP workers must sleep, not spin, or the two pools fight for every core
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters (dense fp32 matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # ibid. (dense bf16 matrix)
HBM_PEAK_GBS = 8000.0          # ibid.
BYTES_PER_SIM = 3200.0         # algorithmic bytes per simulation at 400 sims/move
MLP_FLOP = 2.0 * (70 * 100 + 11 * 100 * 100 + 100 + 100 * 96)

# name -> (kind constant name, architecture, dtype label, matrix peak, MFMA products issued per algorithmic
#          product, kernel name as in the rocprof summaries)
NETS = {
    # (the throughput kernel is pixel-major since round 4: it multiplies only the (pixel, tap) pairs on the board, 6.71 of the
    #  9.65 MFLOP per row that the algorithmic count -- SURVEY 8d, padding included -- holds: 3 x 6.71 / 9.65 products issued)
    "rescnn4h3": ("NET_RESCNN4_H3", "rescnn4", "f32(f16x3)", BF16_MFMA_PEAK_TFLOPS, 2.086, "co_k_rescnn_forward_h3p"),
    "mlp12x100h3": ("NET_MLP12X100_H3", "mlp12x100", "f32(f16x3)", BF16_MFMA_PEAK_TFLOPS, 3.0, "co_k_mlp_forward_h3"),
    "rescnn4x6": ("NET_RESCNN4_X6", "rescnn4", "f32(bf16x6)", BF16_MFMA_PEAK_TFLOPS, 6.0, "co_k_rescnn_forward_x6"),
    "rescnn4": ("NET_RESCNN4", "rescnn4", "f32", FP32_MFMA_PEAK_TFLOPS, 1.0, "co_k_rescnn_forward"),
    "rescnn4x3": ("NET_RESCNN4_X3", "rescnn4", "bf16x3", BF16_MFMA_PEAK_TFLOPS, 3.0, "co_k_rescnn_forward_x3"),
    "mlp12x100x6": ("NET_MLP12X100_X6", "mlp12x100", "f32(bf16x6)", BF16_MFMA_PEAK_TFLOPS, 6.0, "co_k_mlp_forward_x6"),
    "mlp12x100": ("NET_MLP12X100", "mlp12x100", "f32", FP32_MFMA_PEAK_TFLOPS, 1.0, "co_k_mlp_forward"),
    "mlp12x100x3": ("NET_MLP12X100_X3", "mlp12x100", "bf16x3", BF16_MFMA_PEAK_TFLOPS, 3.0, "co_k_mlp_forward_x3"),
}
ARITHMETIC = {
    "f32(f16x3)": "float32 in/out/accumulate; each matrix product as three fp16 MFMA products of two-term operand splits "
                  "(x = fp16(x) + fp16(x - fp16(x)): 22 significand bits per operand, the dropped term is 2^-22 of a product; fp16 "
                  "subnormals kept by converter and matrix pipe): float32-class -- error vs float64 within three times the "
                  "fp32-MFMA kernel's own on every weight set incl. five reference checkpoints (tests/test_net_precision.py, "
                  "tests/test_trained_golden.py), twenty times inside the 1e-4 contract of BASELINE.json",
    "f32(bf16x6)": "float32 in/out/accumulate; each matrix product as six bf16 MFMA products of three-term operand splits "
                   "(the terms sum to the float32 value; dropped terms < 2^-24 of a product): float32-equivalent, error vs "
                   "float64 = the fp32-MFMA kernel's (tests/test_net_precision.py)",
    "f32": "float32 MFMA (v_mfma_f32_16x16x4_f32), a k-ordered float32 fma chain",
    "bf16x3": "two-term bf16 operand splits, three MFMA products: 16 significand bits, narrower than float32",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--spe", type=int, default=16)
    ap.add_argument("--c-puct", type=float, default=1.0)
    ap.add_argument("--epsilon", type=float, default=0.25)
    ap.add_argument("--net", default="rescnn4h3", choices=sorted(NETS),
                    help="rescnn4h3 (default) = the 4-block residual CNN BASELINE.json configs[1] names at float32-class two-term "
                         "fp16 split precision; rescnn4x6 = three-term bf16 split (float32-equivalent); rescnn4 = the same network "
                         "on fp32 MFMA; mlp12x100* = the reference's own net; *x3 = two-term bf16 split, narrower than float32")
    ap.add_argument("--no-mlp-extra", "--no-variants", dest="no_variants", action="store_true",
                    help="skip the other networks reported under detail.variants")
    ap.add_argument("--variant-steps", type=int, default=20, help="timed generations per variant (capped by --steps)")
    ap.add_argument("--stagger", action="store_true", help="keep the reference's staggered start")
    ap.add_argument("--tflite", default="", help="with --net mlp12x100*: weights imported from a reference TFLite checkpoint")
    ap.add_argument("--arena-units", type=int, default=0)
    ap.add_argument("--no-unshared", action="store_true", help="skip the extra single-pool generation behind roofline.unshared (profiling runs)")
    ap.add_argument("--pools", type=int, default=0, help="independent game pools on separate streams per GPU (0 = engine default)")
    ap.add_argument("--cpu-games", type=int, default=64,
                    help="games of the CPU-baseline generation that carries cpu_baseline.value, played to the end (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="safety cap of each CPU-baseline leg (a leg that hits it is reported as extrapolated)")
    ap.add_argument("--recycle-games", type=int, default=16384,
                    help="games of the `recycled` variant: one generation of this many games on --games resident slots "
                         "(a slot whose game ends takes the next game); 0 = skip")
    ap.add_argument("--no-eval-cache", action="store_true",
                    help="evaluate every request row (the engine's evaluation cache off); the default run reports this as the "
                         "`no_eval_cache` variant")
    ap.add_argument("--trained", action="store_true",
                    help="the mlp12x100 kinds search with the reference's last checkpoint (tests/golden/trained_last.npz) instead of "
                         "random-init weights: the narrow, deep trees of every generation after the first (profiling runs)")
    ap.add_argument("--no-trained", action="store_true",
                    help="skip detail.trained_checkpoint (the reference's last checkpoint through the same workload)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip detail.configs (one generation of each other BASELINE configuration: cfg1, cfg4, cfg5, tournament, compat)")
    ap.add_argument("--check-gather", action="store_true",
                    help="with a process group (N > 1, or CORINTHO_FORCE_DIST=1 at N = 1): behind the timed region every rank compares "
                         "its block of one more gathered generation with its own export_samples() byte for byte "
                         "(detail.collectives_per_step.gather_check)")
    ap.add_argument("--engine", default="hip", choices=("hip", "emu"),
                    help="hip = the product (libcorintho_hip.so on an MI355X).  emu = CPU rehearsal of the multi-rank path for "
                         "the tests only: the lane-loop build of the same kernel source (tests/emu) over gloo; never a result")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the
    environment torch.distributed.run would give them), relay rank 0's JSON line, fail if any rank fails.
    Runs before anything touches the GPU in this process.  All children are watched: the first one to fail ends
    the others (a rank that died during set-up would otherwise leave the rest waiting in a collective), and the
    whole job has a time limit.  Children are always fresh processes, never a re-exec of this one."""
    import socket
    import subprocess
    import tempfile

    limit_s = float(os.environ.get("CORINTHO_BENCH_TIMEOUT", "3000"))
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    for attempt in range(3):  # the rendezvous port is free when we look, not necessarily when rank 0 binds it: retry
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_WORLD_SIZE=str(args.gpus), CORINTHO_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        outs = [tempfile.TemporaryFile() for _ in range(args.gpus)]
        errs = [tempfile.TemporaryFile() for _ in range(args.gpus)]
        procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=outs[r], stderr=errs[r])
                 for r in range(args.gpus)]
        t0 = time.time()
        failed = None
        while True:
            rcs = [p.poll() for p in procs]
            bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(rc == 0 for rc in rcs):
                break
            if time.time() - t0 > limit_s:
                failed = -1
                break
            time.sleep(0.2)
        if failed is not None:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
        texts = []
        for f in errs:
            f.seek(0)
            texts.append(f.read().decode(errors="replace"))
        if failed is not None and failed >= 0 and "EADDRINUSE" in texts[failed].upper().replace(" ", "") and attempt < 2:
            continue  # someone took the port between our look and the rank's bind
        for r, t in enumerate(texts):  # what the ranks wrote to stderr is relayed, rank by rank
            if t.strip():
                sys.stderr.write("".join("[rank %d] %s\n" % (r, line) for line in t.rstrip().split("\n")))
        if failed is not None:
            raise SystemExit("bench.py --gpus %d: %s" % (args.gpus, "time limit of %.0f s reached" % limit_s if failed < 0 else
                                                        "rank %d exited with code %s; the other ranks were stopped" % (failed, procs[failed].returncode)))
        outs[0].seek(0)
        sys.stdout.write(outs[0].read().decode())
        sys.stdout.flush()
        return


def measured_traffic(kernel, args, net, npools):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/r06_<net>_pmc.json, else an earlier round's; PMC counters cannot be collected from inside an un-profiled run).
    Only valid for the workload those passes were taken on (the default one); otherwise null."""
    if (args.games, args.sims, args.spe) != (4096, 400, 16):
        return None
    try:
        for rnd in ("r06", "r05", "r04", "r03", "r02"):
            path = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (rnd, net))
            if os.path.exists(path):
                break
        with open(path) as f:
            k = json.load(f)["kernels"]
            t = k[kernel]["traffic_bytes_per_launch"]
            twin = "co_k_rescnn_forward_h3_small" if kernel == "co_k_rescnn_forward_h3p" else kernel + "_small"
            if twin in k:
                # a network launch queues the small-batch instance always and the throughput instance when the batch can
                # exceed the former's rows; one of them works: bytes of both over the number of network launches
                n, nt = k[kernel].get("launches", 1), k[twin].get("launches", 1)
                t = (t * n + k[twin]["traffic_bytes_per_launch"] * nt) / max(n, nt, 1)
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


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_leg(args, G, threads, net_name, weights, cap_s, evals_per_game):
    """One leg of the CPU baseline: the oracle (CPU restatement of the reference's OpenMP path,
    trainer.cpp:175-196) on `threads` host threads plays ONE WHOLE generation of `G` games of the bench
    workload, the network evaluated on the CPU between iterations as the reference's Keras loop does
    (main.pyx:70-83).  games/s = G / wall time of the finished generation.  `cap_s` is a safety net only: a
    leg still running then stops and reports its rate scaled by the evaluations a game needs
    (finished = false)."""
    from corintho_ai_amd import nets
    from tests import ref_nets
    from oracle import oracle as O
    from tests import harness as H

    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=threads)  # numpy's BLAS on the same threads
    except Exception:
        pass
    if net_name == "rescnn4":
        import torch

        torch.set_num_threads(threads)
        fwd = lambda s: ref_nets.rescnn4_forward_ref(weights, s)  # noqa: E731
    elif net_name == "mlp12x100":
        fwd = lambda s: ref_nets.mlp12x100_forward_np(weights, s)  # noqa: E731
    else:
        fwd = H.uniform_net  # zero-cost stand-in: the search alone
    t = O.Trainer(G, seed=12345, max_searches=args.sims, searches_per_eval=args.spe, c_puct=args.c_puct,
                  epsilon=args.epsilon, num_threads=threads)
    t.set_stagger(False)  # as the GPU run
    evals = np.zeros(G * args.spe, np.float32)
    probs = np.zeros((G * args.spe, 96), np.float32)
    states = np.zeros((G * args.spe, 70), np.float32)
    rows = 0
    nn_s = 0.0
    iters = 0
    t0 = time.perf_counter()
    done = False
    while not done:
        done = t.doIteration(evals, probs, -1)
        if done:
            break
        n = t.num_requests(-1)
        t.writeRequests(states, -1)
        t1 = time.perf_counter()
        e, p = fwd(states[:n])
        if net_name != "none":
            nn_s += time.perf_counter() - t1
        evals[:n] = e
        probs[:n] = p
        rows += n
        iters += 1
        if time.perf_counter() - t0 >= cap_s:
            break
    dt = time.perf_counter() - t0
    return {"games_per_s": G / dt if done else rows / dt / evals_per_game, "leaf_evals_per_s": rows / dt, "threads": threads,
            "net": net_name, "games": G, "seconds": dt, "network_seconds": nn_s, "iterations": iters, "rows": rows,
            "finished": bool(done)}


def cpu_baseline(args, arch, weights_by_arch, evals_per_game):
    """Whole (small) generations of the bench workload on the host cores, one per leg, sized so that each takes
    seconds: `--cpu-games` games for the leg that carries `value` (all cores, the bench's architecture), fewer for
    one thread, more for the cheap network and for the search alone."""
    cores = host_cores()
    G = args.cpu_games
    legs = {}
    other = "mlp12x100" if arch == "rescnn4" else "rescnn4"
    size = {"rescnn4": (G, max(G // 4, 8)), "mlp12x100": (4 * G, 2 * G), "none": (16 * G, 4 * G)}
    for name, threads, net in (("all_cores", cores, arch), ("one_thread", 1, arch),
                               ("all_cores_" + other, cores, other), ("one_thread_" + other, 1, other),
                               ("all_cores_search_only", cores, "none"), ("one_thread_search_only", 1, "none")):
        legs[name] = cpu_leg(args, size[net][0 if threads > 1 else 1], threads, net, weights_by_arch.get(net), args.cpu_seconds, evals_per_game)
    main = legs["all_cores"]
    return {
        "value": main["games_per_s"],
        "unit": "games/s",
        "cores": cores,
        "kind": "port",
        "extrapolated": not main["finished"],
        "cpu_model": cpu_model(),
        "sample": "oracle/ (C restatement of trainer.cpp/selfplayer.cpp/trainmc.cpp, OpenMP over games) + float32 %s on the host "
                  "(%s), %d threads: one whole generation of %d games of the bench workload (%d sims/move, spe %d), played to the "
                  "end%s: %d iterations, %d request rows, %.1f s, %.1f s of it network"
                  % (arch, "torch CPU" if arch == "rescnn4" else "numpy", cores, main["games"], args.sims, args.spe,
                     "" if main["finished"] else " -- NOT reached within the cap, rate scaled by evaluations per game", main["iterations"],
                     main["rows"], main["seconds"], main["network_seconds"]),
        "one_thread": legs["one_thread"]["games_per_s"],
        "legs": legs,
    }


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args)  # N child processes, one per GPU; this process never touches a GPU
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %s" % (args.gpus, os.environ["WORLD_SIZE"]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    emu = args.engine == "emu"
    if os.environ.get("CORINTHO_BENCH_TEST_FAIL_RANK") == str(rank):  # tests/test_distributed_cpu.py: a rank that dies during set-up
        raise SystemExit(3)
    dist = None
    torch = None
    use_dist = world > 1 or os.environ.get("CORINTHO_FORCE_DIST") == "1"  # the latter: 1-rank rehearsal of the RCCL path
    dev = "cpu" if emu else "cuda"
    if use_dist:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL (and gloo) print a banner on stdout when the communicator comes up; stdout carries ONE JSON line:
        # the communicator is brought up (first collective) with stdout pointing at stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if emu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
                dist.all_reduce(torch.zeros(1))
            else:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                warm = torch.zeros(1, device="cuda")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import corintho_ai_amd as CA
    from corintho_ai_amd import Trainer, nets

    cdll = None
    if emu:
        from tests.emu import emulib

        cdll = emulib.load()

    G = args.games
    weights_by_arch = {"rescnn4": nets.init_rescnn4(0), "mlp12x100": nets.init_mlp12x100(0)}
    if args.tflite:
        from corintho_ai_amd.tflite_import import mlp12x100_from_tflite

        weights_by_arch["mlp12x100"] = mlp12x100_from_tflite(args.tflite)
    if args.trained:
        weights_by_arch["mlp12x100"] = np.load(os.path.join(ROOT, "tests", "golden", "trained_last.npz"))["weights"]
    flop_by_arch = {"rescnn4": nets.rescnn4_flop_per_row(), "mlp12x100": MLP_FLOP}

    def make_trainer(pools, games=None, resident=-1, eval_cache=None):
        n = G if games is None else games
        return Trainer(n, "", 12345, args.sims, args.spe, args.c_puct, args.epsilon, 0, 1, False, device=local_rank,
                       stagger=args.stagger, arena_units=args.arena_units, game_base=rank * n, total_games=world * n,
                       pools=pools, resident=resident, eval_cache=(not args.no_eval_cache) if eval_cache is None else eval_cache,
                       _cdll=cdll)

    tr = make_trainer(args.pools)
    gatherer = None
    if use_dist:
        from corintho_ai_amd.dist import SampleGather

        gatherer = SampleGather(tr, G, on_device=not emu)

    def batch_fill(totals, slots):
        """rows per network launch against what a launch can hold (one pool's slots x searches per evaluation)"""
        pools_ = max(int(totals.get("pools", 1)), 1)
        cap = slots * args.spe / pools_
        rows = totals["nn_rows"] / max(totals["nn_launches"], 1)
        return {"rows_per_launch": rows, "capacity_rows": cap, "fill": rows / cap,
                "rows_evaluated_per_launch": totals["nn_rows_evaluated"] / max(totals["nn_launches"], 1)}

    STAT_KEYS = ("searches", "evals", "plies", "iterations", "mcts_ms", "nn_ms", "pack_ms", "nn_rows", "nn_rows_evaluated", "nn_launches",
                 "mcts_launches", "timed_launches", "nn_timed_rows", "mcts_timed_ms", "nn_timed_ms", "steps_cut")

    def barrier():
        if use_dist:
            dist.barrier()
            if not emu:
                torch.cuda.synchronize()

    def run_generations(trainer, net, steps, warmup, seed0, collective, weights=None):
        """`warmup` untimed + `steps` timed generations of `net`; -> (seconds, totals)"""
        kind_name, arch = NETS[net][0], NETS[net][1]
        trainer.set_net(getattr(CA, kind_name), weights_by_arch[arch] if weights is None else weights)
        totals = dict.fromkeys(STAT_KEYS, 0)
        totals.update(gather_ms=0.0, samples=0, peak_arena_units=0, pools=1, gather_bytes=0, score=0.0, unfinished=0,
                      gathered_samples=0)

        def one(seed, timed):
            trainer.reset(seed)
            done = trainer.run()
            if collective:
                # every rank enters both collectives whatever its own generation did: a rank that raised before
                # them would leave the others waiting in the all-gather
                t0 = time.perf_counter()
                counts, _, _ = gatherer.gather()  # C1: counts + one all-gather of the un-augmented samples
                score, unfinished = gatherer.score_and_unfinished(done)  # C2: one all-reduce of two scalars
                if timed:
                    totals["gather_ms"] += (time.perf_counter() - t0) * 1e3
                    totals["gather_bytes"] += gatherer.bytes_moved
                    totals["score"] += score
                    totals["unfinished"] += unfinished
                    totals["gathered_samples"] += int(counts.sum())
                if unfinished:
                    raise RuntimeError("generation did not finish on %d rank(s)" % unfinished)
            elif not done:
                raise RuntimeError("generation did not finish")
            if timed:
                st = trainer.stats()
                for k in STAT_KEYS:
                    totals[k] += st.get(k, 0)
                totals["step_budget_last"] = st.get("step_budget_last", 0)
                totals["pools"] = st["pools"]
                totals["samples"] += trainer.num_samples()
                totals["peak_arena_units"] = max(totals["peak_arena_units"], st["peak_arena_units"])

        for w in range(warmup):
            one(1000 + seed0 + w, False)
        barrier()
        t0 = time.perf_counter()
        for s in range(steps):
            one(seed0 + s, True)
        barrier()
        return time.perf_counter() - t0, totals

    def rooflines_of(net, totals, npools, wall_s=None, generations=1):
        """roofline objects of both kernel families of a measured run: (dominant, {"network": .., "search": ..}).
        Per-launch rates from the HIP-event durations of the timed launches on the pools' streams (one iteration per
        pool and window of eight carries events); `wall` = the same algorithmic work over the run's wall time."""
        _, arch, dtype, peak, issued, kname = NETS[net]
        flop_per_row = flop_by_arch[arch]
        nn_s, mcts_s = totals["nn_ms"] * 1e-3, totals["mcts_ms"] * 1e-3
        tl = totals["timed_launches"]
        if tl > 0:
            achieved = totals["nn_timed_rows"] * flop_per_row / max(totals["nn_timed_ms"] * 1e-3, 1e-12) / 1e12
        else:
            achieved = totals["nn_rows_evaluated"] * flop_per_row / max(nn_s, 1e-12) / 1e12
        useful = nets.rescnn4_useful_flop_per_row() / flop_per_row if arch == "rescnn4" else 1.0
        rn = {"kernel": kname, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
              "frac": achieved / peak, "traffic": measured_traffic(kname, args, net, npools),
              "issued_frac": issued * achieved / peak,
              # products of a 3x3 tap with the zero padding of the 4x4 board (44 of 144) not counted as work
              "algorithmic_useful": {"achieved": achieved * useful, "frac": achieved * useful / peak},
              "algorithmic": "%.1f KFLOP/row x %d rows in %d timed launches of %d (x%.2f MFMA products issued per algorithmic one)"
                             % (flop_per_row / 1e3, totals["nn_timed_rows"] if tl else totals["nn_rows_evaluated"],
                                tl if tl else totals["nn_launches"], totals["nn_launches"], issued),
              "avg_launch_ms": (totals["nn_timed_ms"] / tl) if tl else totals["nn_ms"] / max(totals["nn_launches"], 1)}
        sims_per_launch = totals["searches"] / max(totals["mcts_launches"], 1)
        s_launch_ms = (totals["mcts_timed_ms"] / tl) if tl else totals["mcts_ms"] / max(totals["mcts_launches"], 1)
        achieved_s = sims_per_launch * BYTES_PER_SIM / max(s_launch_ms * 1e-3, 1e-12) / 1e9
        rs = {"kernel": "co_k_mcts_step", "bound": "hbm", "achieved": achieved_s, "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": achieved_s / HBM_PEAK_GBS,
              "traffic": measured_traffic("co_k_mcts_step", args, net, npools),
              "algorithmic": "%.0f B/simulation x %.0f simulations per launch (%d simulations in %d launches)"
                             % (BYTES_PER_SIM, sims_per_launch, totals["searches"], totals["mcts_launches"]),
              "avg_launch_ms": s_launch_ms,
              "note": "graph work classified under the HBM roofline (bytes per simulation, SURVEY 8d); what bounds it is one "
                      "wavefront's dependent instruction chain per game, not bandwidth (DESIGN section 6)"}
        if wall_s:
            # the same algorithmic work over the wall time of the timed region: <= 1 by construction (the per-launch rates
            # of two pools' streams overlap in time and may add up to more than the wall-level one)
            rn["wall"] = {"achieved": totals["nn_rows_evaluated"] * flop_per_row / wall_s / 1e12,
                          "frac": totals["nn_rows_evaluated"] * flop_per_row / wall_s / 1e12 / peak}
            rs["wall"] = {"achieved": totals["searches"] * BYTES_PER_SIM / wall_s / 1e9,
                          "frac": totals["searches"] * BYTES_PER_SIM / wall_s / 1e9 / HBM_PEAK_GBS}
        rn["streams"] = rs["streams"] = npools
        return (rn if nn_s >= mcts_s else rs), {"network": rn, "search": rs}

    def roofline_of(net, totals, npools, wall_s=None):
        return rooflines_of(net, totals, npools, wall_s)[0]

    # ------------------------------------------------------------------ the timed region
    dt, totals = run_generations(tr, args.net, args.steps, args.warmup, 0, use_dist)
    per_rank = [G * args.steps / dt]
    ranks_seen = 1
    if use_dist:
        mine = torch.tensor([dt], dtype=torch.float64, device=dev)
        every = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, mine)
        per_rank = [G * args.steps / float(x) for x in every.tolist()]
        ranks_seen = len(per_rank)
        dt = float(every.max().item())  # the job takes as long as its slowest rank
        agg = torch.tensor([totals["searches"], totals["evals"], totals["nn_rows"], totals["samples"]], dtype=torch.float64,
                           device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        job_searches, job_evals, job_rows, job_samples = [float(x) for x in agg.tolist()]
    else:
        job_searches, job_evals, job_rows = float(totals["searches"]), float(totals["evals"]), float(totals["nn_rows"])
        job_samples = float(totals["samples"])

    gather_check = None
    if use_dist and args.check_gather:
        # Trainer::writeSamples order over the whole job (trainer.cpp:103-113): rank r's rows stand behind those of the
        # ranks before it; each rank holds the bytes it contributed, so each checks its own block of what came back
        tr.reset(424242)
        assert tr.run()
        t0 = time.perf_counter()
        sp_all, oc_all = gatherer.rows()
        g_ms = (time.perf_counter() - t0) * 1e3
        counts = gatherer.all_cnt.cpu().numpy().astype(np.int64)
        lo = int(counts[:rank].sum())
        sp_own, oc_own = tr.export_samples()
        equal = (sp_all.shape[0] == int(counts.sum()) and int(counts[rank]) == sp_own.shape[0] == tr.num_samples()
                 and sp_all[lo:lo + sp_own.shape[0]].tobytes() == sp_own.tobytes()
                 and oc_all[lo:lo + oc_own.shape[0]].tobytes() == oc_own.tobytes())
        flag = torch.tensor([0.0 if equal else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(flag)
        gather_check = {"rows_gathered": int(sp_all.shape[0]), "rows_own": int(sp_own.shape[0]), "num_samples": tr.num_samples(),
                        "ranks_with_a_difference": int(flag.item()), "bytes_equal_to_export_samples": int(flag.item()) == 0,
                        "gather_and_copy_back_ms": g_ms, "payload_on": gatherer.dev, "backend": dist.get_backend(),
                        "payload_bytes": gatherer.bytes_moved}

    if rank == 0:
        _, arch, dtype, peak, issued, kname = NETS[args.net]
        flop_per_row = flop_by_arch[arch]
        games_total = world * G * args.steps
        value = games_total / dt
        nn_s = totals["nn_ms"] * 1e-3
        mcts_s = totals["mcts_ms"] * 1e-3
        npools = int(totals.get("pools", 1))
        roofline, both = rooflines_of(args.net, totals, npools, wall_s=dt)
        if npools > 1 and not args.no_unshared and world == 1:
            # The timed region runs the games as `npools` pools on separate streams, so the durations
            # above are those of kernels SHARING the GPU with the other pool's kernels (their sum
            # exceeds the wall time).  One more generation with a single pool gives the same kernel's
            # rate when it has the GPU to itself.
            t1 = make_trainer(1)
            du, tu = run_generations(t1, args.net, 1, 1, 500, False)
            r1 = roofline_of(args.net, tu, 1)
            roofline["unshared"] = {"achieved": r1["achieved"], "frac": r1["frac"], "issued_frac": r1.get("issued_frac"),
                                    "avg_launch_ms": r1["avg_launch_ms"], "games_per_s_single_pool": G / du,
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
            "dtype": dtype,
            "data": "synthetic",
            "config": {
                "workload": "%d parallel self-play games per GPU, %d sims/move, %d searches/eval, %s (%s), "
                            "fused on-device search + inference, %d pool(s) per GPU, %s"
                            % (G, args.sims, args.spe, args.net,
                               "the reference's last checkpoint" if args.trained and "mlp" in args.net else "random init, seed 0", npools,
                               "staggered start" if args.stagger else "no stagger"),
                "games_per_gpu": G, "sims_per_move": args.sims, "searches_per_eval": args.spe, "net": args.net,
                "arithmetic": ARITHMETIC[dtype],
                "c_puct": args.c_puct, "epsilon": args.epsilon, "parallelism": "games sharded x%d" % world,
            },
            "roofline": roofline,
            # the other kernel family of the same run (BASELINE.md section 3 asks for the search kernel AND the inference kernel)
            "roofline_search": both["search"],
            "roofline_network": both["network"],
            "detail": {
                "sims_per_s": job_searches / dt,
                "leaf_evals_per_s": job_evals / dt,
                "plies_per_game": totals["plies"] / max(G * args.steps, 1),
                "evals_per_game": totals["evals"] / max(G * args.steps, 1),
                "iterations_per_step": totals["iterations"] / max(args.steps, 1),
                "rank0_device_ms_per_step": {"mcts": totals["mcts_ms"] / args.steps, "network": totals["nn_ms"] / args.steps,
                                             "pack": totals["pack_ms"] / args.steps,
                                             "sample_gather_and_score_allreduce": totals["gather_ms"] / args.steps},
                "mcts_GBps_algorithmic": totals["searches"] * BYTES_PER_SIM / max(mcts_s, 1e-12) / 1e9,
                "network_TFLOPs_algorithmic": totals["nn_rows_evaluated"] * flop_per_row / max(nn_s, 1e-12) / 1e12,
                "nn_rows_requested": totals["nn_rows"], "nn_rows_evaluated": totals["nn_rows_evaluated"],
                "nn_rows_evaluated_over_requested": totals["nn_rows_evaluated"] / max(totals["nn_rows"], 1),
                "evaluation_cache": "off" if args.no_eval_cache else
                                    "on: a request row whose position was evaluated earlier in the SAME generation receives the stored "
                                    "outputs (bit-identical: a row's outputs depend on the row only); emptied at every generation start",
                "peak_arena_units_per_tree": totals["peak_arena_units"],
                "network_batch": batch_fill(totals, G),
                # ca_config.step_budget (automatic): game steps that stopped at their time budget and went on in the next iteration
                "step_budget": {"steps_cut_per_step": totals["steps_cut"] / max(args.steps, 1),
                                "game_steps_per_step": totals["evals"] / max(args.steps, 1) / args.spe,
                                "last_budget_us": totals.get("step_budget_last", 0),
                                "note": "a step past 1.5 x the running mean of its pool's steps holds its queued leaves back and resumes in the "
                                        "next launch: identical per-game results, more and shorter iterations"},
                "world_size": dist.get_world_size() if use_dist else 1,
                "ranks_seen": ranks_seen,
                "per_rank_games_per_s": per_rank,
                "launched_by": "bench.py itself (one child process per GPU)" if os.environ.get("CORINTHO_BENCH_SELF_LAUNCHED")
                               else ("torch.distributed.run / the caller's environment" if use_dist else "single process"),
            },
        }
        if use_dist:
            out["detail"]["collectives_per_step"] = {
                "all_gather_counts": 1, "all_gather_samples": 1, "all_reduce_score_done": 1,
                "sample_bytes_gathered_per_step": totals["gather_bytes"] / max(args.steps, 1),
                "generation_score": totals["score"] / max(args.steps, 1), "unfinished_games": totals["unfinished"],
                "samples_gathered": totals["gathered_samples"], "samples_of_all_shards": job_samples}
            if gather_check is not None:
                out["detail"]["collectives_per_step"]["gather_check"] = gather_check
        if world == 1 and not args.no_variants:
            # the other network kinds on the same pool: the same timed loop, each with its own roofline object
            vsteps = max(1, min(args.variant_steps, args.steps))
            variants = {}
            for name in ("rescnn4x6", "rescnn4", "rescnn4x3", "mlp12x100h3", "mlp12x100x6", "mlp12x100", "mlp12x100x3", "rescnn4h3"):
                if name == args.net:
                    continue
                d1, t1 = run_generations(tr, name, vsteps, 1, 7000, False)
                vflop = flop_by_arch[NETS[name][1]]
                variants[name] = {"games_per_s": G * vsteps / d1, "ms_per_step": d1 * 1e3 / vsteps, "steps": vsteps, "warmup": 1,
                                  "dtype": NETS[name][2], "roofline": roofline_of(name, t1, int(t1.get("pools", 1)), wall_s=d1),
                                  "network_TFLOPs_algorithmic": t1["nn_rows_evaluated"] * vflop / max(t1["nn_ms"] * 1e-3, 1e-12) / 1e12,
                                  "nn_rows_evaluated_over_requested": t1["nn_rows_evaluated"] / max(t1["nn_rows"], 1),
                                  "mcts_GBps_algorithmic": t1["searches"] * BYTES_PER_SIM / max(t1["mcts_ms"] * 1e-3, 1e-12) / 1e9,
                                  "device_ms_per_step": {"mcts": t1["mcts_ms"] / vsteps, "network": t1["nn_ms"] / vsteps}}
            out["detail"]["variants"] = variants
            if not args.no_eval_cache:
                # the same workload with every request row evaluated (the reference's own behaviour, main.pyx:70-83)
                tr3 = make_trainer(args.pools, eval_cache=False)
                d3, t3 = run_generations(tr3, args.net, min(5, args.steps), 1, 11000, False)
                n3 = min(5, args.steps)
                out["detail"]["no_eval_cache"] = {"games_per_s": G * n3 / d3, "ms_per_step": d3 * 1e3 / n3, "steps": n3, "warmup": 1,
                                                  "roofline": roofline_of(args.net, t3, int(t3.get("pools", 1)), wall_s=d3),
                                                  "nn_rows_evaluated_over_requested": t3["nn_rows_evaluated"] / max(t3["nn_rows"], 1)}
                del tr3
        if world == 1 and not args.no_variants and npools != 2:
            # rounds 1-4 ran two pools per GPU; the same workload that way, for continuity of the per-launch figures
            trp = make_trainer(2)
            n2 = min(5, args.steps)
            dp, tp = run_generations(trp, args.net, n2, 1, 13000, False)
            out["detail"]["two_pools"] = {"games_per_s": G * n2 / dp, "ms_per_step": dp * 1e3 / n2, "steps": n2, "warmup": 1,
                                          "roofline": roofline_of(args.net, tp, int(tp.get("pools", 1)), wall_s=dp)}
            del trp
        if world == 1 and args.recycle_games > G:
            # one generation of `recycle_games` games on G resident slots: a slot whose game ends takes the next game
            # (ca_config.resident), so the launches stay full until the games run out instead of thinning with the
            # generation's longest games
            tr2 = make_trainer(args.pools, games=args.recycle_games, resident=G)
            rsteps = max(1, min(3, args.steps))
            d2, t2 = run_generations(tr2, args.net, rsteps, 1, 9000, False)
            out["detail"]["recycled"] = {
                "games_per_s": args.recycle_games * rsteps / d2, "games": args.recycle_games, "resident_slots": G,
                "steps": rsteps, "warmup": 1, "ms_per_step": d2 * 1e3 / rsteps,
                "network_batch": batch_fill(t2, G), "roofline": roofline_of(args.net, t2, int(t2.get("pools", 1)), wall_s=d2),
                "iterations_per_step": t2["iterations"] / rsteps,
                "note": "same workload per game; %d games per generation on %d slots" % (args.recycle_games, G)}
            del tr2
        trained_path = os.path.join(ROOT, "tests", "golden", "trained_last.npz")
        if world == 1 and not args.no_trained and not emu and os.path.exists(trained_path):
            # The regime every generation after the first lives in (main.pyx:285-322): the search guided by a TRAINED
            # network -- narrow, deep trees, in late plies every second simulation terminal.  The reference's own
            # architecture with the reference's last checkpoint (rating/tflite_models, imported once by
            # tools/gen_trained_golden.py and committed as data), same 4096 x 400 workload, f16x3 arithmetic.
            wt = np.load(trained_path)["weights"]
            nt = min(5, args.steps)
            dtc, ttc = run_generations(tr, "mlp12x100h3", nt, 1, 15000, False, weights=wt)
            rt_dom, rt_both = rooflines_of("mlp12x100h3", ttc, int(ttc.get("pools", 1)), wall_s=dtc)
            for r_ in (rt_both["search"], rt_both["network"]):  # the counters of THIS workload (profiles/r06_mlp12x100h3_trained_pmc.json)
                r_["traffic"] = measured_traffic(r_["kernel"], args, "mlp12x100h3_trained", int(ttc.get("pools", 1)))
            out["detail"]["trained_checkpoint"] = {
                "net": "mlp12x100h3", "weights": "tests/golden/trained_last.npz (the reference's last TFLite checkpoint)",
                "games_per_s": G * nt / dtc, "ms_per_step": dtc * 1e3 / nt, "steps": nt, "warmup": 1,
                "plies_per_game": ttc["plies"] / max(G * nt, 1), "evals_per_game": ttc["evals"] / max(G * nt, 1),
                "sims_per_s": ttc["searches"] / dtc, "iterations_per_step": ttc["iterations"] / nt,
                "steps_cut_per_step": ttc["steps_cut"] / nt, "last_budget_us": ttc.get("step_budget_last", 0),
                "device_ms_per_step": {"mcts": ttc["mcts_ms"] / nt, "network": ttc["nn_ms"] / nt},
                "roofline": rt_dom, "roofline_search": rt_both["search"], "roofline_network": rt_both["network"]}
        if world == 1 and not args.no_configs and not emu and (G, args.sims, args.spe) == (4096, 400, 16):
            # BASELINE.json's other configurations, one generation each on the final tree (tools/run_configs.py), each with
            # its roofline objects: cfg1 (64 x 50), cfg4 (4096 x 1600 + Dirichlet noise: toml/train.toml:2-18), cfg5 (arena,
            # main.pyx:329-349), a tournament, and the host-driven reference protocol (PCIe-inclusive, never `value`)
            tr.close()  # cfg4's trees take 118 GB: give the default pool's memory back first
            from tools import run_configs as RC

            out["detail"]["configs"] = RC.measure_all(device=local_rank)
        # the numbers a reader compares `value` with, in one place (each is measured above, same workload)
        d = out["detail"]
        out["detail"]["summary"] = {
            "games_per_s": value,
            "no_eval_cache_games_per_s": d.get("no_eval_cache", {}).get("games_per_s"),
            "float32_equivalent_bf16x6_games_per_s": d.get("variants", {}).get("rescnn4x6", {}).get("games_per_s"),
            "fp32_mfma_games_per_s": d.get("variants", {}).get("rescnn4", {}).get("games_per_s"),
            "reference_network_mlp12x100h3_games_per_s": d.get("variants", {}).get("mlp12x100h3", {}).get("games_per_s"),
            "recycled_games_per_s": d.get("recycled", {}).get("games_per_s"),
            "two_pools_games_per_s": d.get("two_pools", {}).get("games_per_s"),
            "trained_checkpoint_mlp12x100h3_games_per_s": d.get("trained_checkpoint", {}).get("games_per_s"),
            "configs_games_per_s": {k: v.get("games_per_s", v.get("matches_per_s")) for k, v in d.get("configs", {}).items()},
        }
        if world == 1 and args.cpu_games > 0:
            out["cpu_baseline"] = cpu_baseline(args, arch, weights_by_arch, out["detail"]["evals_per_game"])
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
