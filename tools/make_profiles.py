#!/usr/bin/env python3
"""Turns the rocprofv3 results of tools/profile_round.sh (rocpd sqlite files under
gpurun_out/prof_<tag>_{stats,fetch,write,sq}) into the committed summaries:
profiles/<tag>_kernel_stats.md and profiles/<tag>_pmc.json.
usage: make_profiles.py <tag> "<command that was profiled>" [results-dir] [output-dir] """
import collections
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cmd = sys.argv[1], sys.argv[2]
SRC = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out")
DST = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles")


def kname(name):
    """kernel name as the docs use it: no signature, the instances of the network templates under their macro names"""
    n = name.split("(")[0].replace("void ", "").strip()
    for a, b in (("co_k_rescnn_forward_split_t<1, 3>", "co_k_rescnn_forward_x6"), ("co_k_rescnn_forward_split_t<2, 2>", "co_k_rescnn_forward_x3"),
                 ("co_k_rescnn_forward_split_t<1, 2>", "co_k_rescnn_forward_x3_small"), ("co_k_mlp_forward_split_t<3>", "co_k_mlp_forward_x6"),
                 ("co_k_mlp_forward_split_t<2>", "co_k_mlp_forward_x3"), ("co_k_mlp_forward_split_t<3, false>", "co_k_mlp_forward_x6"),
                 ("co_k_mlp_forward_split_t<2, false>", "co_k_mlp_forward_x3"), ("co_k_mlp_forward_split_t<2, true>", "co_k_mlp_forward_h3")):
        n = n.replace(a, b)
    return n


def db_of(kind):
    f = glob.glob(os.path.join(SRC, "prof_%s_%s" % (tag, kind), "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def counters(db):
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for k, did, c, v in db.execute("select %s, dispatch_id, counter_name, value from counters_collection" % kcol):
        k = kname(k)
        acc[k][c] += v
        disp[(k, c)].add(did)
    return {k: {c: (acc[k][c] / max(len(disp[(k, c)]), 1), len(disp[(k, c)])) for c in acc[k]} for k in acc}


lines = ["# %s: kernel statistics (rocprofv3 --kernel-trace --stats), MI355X gfx950, ROCm 7.2" % tag, "",
         "Command: `%s`" % cmd, "", "| kernel | calls | total (us) | average (us) | share % |", "|---|---:|---:|---:|---:|"]
db = db_of("stats")
rows = {}
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    lines.append("| `%s` | %d | %.1f | %.3f | %.2f |" % (kname(name), calls, total, avg, pct))
    rows[kname(name)] = (calls, total, pct)
# a network launch queues two instances of some kernels (the row count on the device picks the one that works, the other
# returns at once): one row per pair = what bench.py times as one launch
for main, twin in (("co_k_rescnn_forward_x3", "co_k_rescnn_forward_x3_small"), ("co_k_rescnn_forward_h3", "co_k_rescnn_forward_h3_small"),
                   ("co_k_rescnn_forward_h3p", "co_k_rescnn_forward_h3_small")):
    if main in rows and twin in rows:
        # (since round 4 a launch whose batch cannot exceed the small-batch kernel's rows does not queue the throughput
        # kernel at all: the network launches are the calls of the kernel that is always queued)
        calls, total, pct = max(rows[main][0], rows[twin][0]), rows[main][1] + rows[twin][1], rows[main][2] + rows[twin][2]
        lines.append("| `%s` + `%s` (one network launch) | %d | %.1f | %.3f | %.2f |" % (main, twin.replace(main, ""), calls, total, total / max(calls, 1), pct))
pmc = {"tag": tag, "command": cmd,
       "note": "rocprofv3 --pmc passes, one counter group per pass with --kernel-trace only. FETCH_SIZE / WRITE_SIZE "
               "are in KB (rocprofv3 units); HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes "
               "(gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64-byte requests as 32).",
       "kernels": {}}
f, w = db_of("fetch"), db_of("write")
if f and w:
    cf, cw = counters(f), counters(w)
    for k in cf:
        if k in cw and "FETCH_SIZE" in cf[k] and "WRITE_SIZE" in cw[k]:
            fe, n = cf[k]["FETCH_SIZE"]
            wr, _ = cw[k]["WRITE_SIZE"]
            pmc["kernels"][k] = {"launches": n, "FETCH_SIZE_KB_per_launch": fe, "WRITE_SIZE_KB_per_launch": wr,
                                 "traffic_bytes_per_launch": (2.0 * fe + wr) * 1024.0}
sq = db_of("sq")
if sq:
    lines += ["", "SQ counters per launch (own pass):", "", "| kernel | counter | per launch |", "|---|---|---:|"]
    for k, d in counters(sq).items():
        if not k.startswith("co_k_") or k in ("co_k_scan", "co_k_compact"):
            continue
        for c in sorted(d):
            lines.append("| `%s` | %s | %.0f |" % (k, c, d[c][0]))
        pmc["kernels"].setdefault(k, {})["sq"] = {c: d[c][0] for c in d}
    # the ratios DESIGN.md quotes, so that they follow from this file alone: of a kernel's wave cycles, the
    # share spent at s_waitcnt, the share with a vector instruction issuing, and the matrix pipe's busy share.
    # SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD cycles with the pipe busy; SQ_BUSY_CYCLES per SE x4 -> the same normalisation
    # as the microarchitecture guide's: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x CUs-per-SE share); given the
    # guide's caveat about that counter on gfx950 the per-wave form is printed as well (busy cycles / wave cycles x waves per SIMD).
    lines += ["", "Derived (same pass): share of the wave cycles waiting (SQ_WAIT_ANY), issuing any instruction (SQ_ACTIVE_INST_ANY), "
              "issuing vector instructions (SQ_ACTIVE_INST_VALU); matrix-pipe cycles per wave cycle (SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES: x waves "
              "per SIMD = the pipe's busy share while the kernel's waves are resident):", "",
              "| kernel | wait % | any instruction % | VALU % | MFMA busy cycles / wave cycles |", "|---|---:|---:|---:|---:|"]
    for k, d in counters(sq).items():
        if not k.startswith("co_k_") or k in ("co_k_scan", "co_k_compact") or "SQ_WAVE_CYCLES" not in d:
            continue
        wc = max(d["SQ_WAVE_CYCLES"][0], 1.0)
        g = lambda c: d[c][0] if c in d else float("nan")  # noqa: E731
        lines.append("| `%s` | %.1f | %.1f | %.1f | %.3f |" % (k, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc,
                                                              100 * g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_VALU_MFMA_BUSY_CYCLES") / wc))
        pmc["kernels"][k]["derived"] = {"wait_share": g("SQ_WAIT_ANY") / wc, "any_inst_share": g("SQ_ACTIVE_INST_ANY") / wc,
                                        "valu_share": g("SQ_ACTIVE_INST_VALU") / wc, "mfma_busy_per_wave_cycle": g("SQ_VALU_MFMA_BUSY_CYCLES") / wc}
os.makedirs(DST, exist_ok=True)
with open(os.path.join(DST, "%s_kernel_stats.md" % tag), "w") as fh:
    fh.write("\n".join(lines) + "\n")
with open(os.path.join(DST, "%s_pmc.json" % tag), "w") as fh:
    json.dump(pmc, fh, indent=1)
print("\n".join(lines))
print(json.dumps(pmc["kernels"], indent=1)[:1500])
