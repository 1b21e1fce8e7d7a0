#!/usr/bin/env python
"""Turn the rocprofv3 outputs of scripts/profile_round.sh (rocpd SQLite databases) into the
committed summaries under profiles/: per-kernel launch statistics of the timed bench run, and
per-launch HBM-side traffic of the path's kernels from the TCC_EA0 request counters."""
import glob
import json
import sqlite3
import sys
from collections import OrderedDict, defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
SHORT = OrderedDict([("k_raycast_batched", "raycast"), ("k_integrate_listed_rest", "integrate_rest"),
                     ("k_integrate_listed<true>", "integrate_bg"), ("k_integrate_listed<(bool)1>", "integrate_bg"),
                     ("k_integrate_listed", "integrate"), ("k_integrate_cull", "integrate_cull"),
                     ("k_far_bounds_listed", "far_bounds_listed"), ("k_far_bounds", "far_bounds_scan"), ("k_far_init", "far_init"), ("k_sign_maps", "sign_maps"),
                     ("k_relevant_tiles", "relevant_tiles"), ("k_relevant_reset", "relevant_reset"),
                     ("k_integrate_batched", "integrate"),
                     ("k_estep", "assoc"), ("k_composite", "composite"), ("k_vis_counts", "vis_counts"),
                     ("k_vis_flags", "vis_flags"), ("k_dilate_batched", "dilate_flags"),
                     ("k_compute_points", "points"), ("k_update_fgbg", "fgbg"), ("k_fg_probs", "fg_probs"),
                     ("k_occluded", "occluded"), ("fillBuffer", "memset"), ("copyBuffer", "memcpy")])


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return name[:40]


def db(path):
    f = glob.glob(f"{src}/{path}/*.db")
    return sqlite3.connect(f[0]) if f else None


out_md = [f"# rocprofv3 summary, round tag `{tag}`", "",
          "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 60 --warmup 30 "
          "--no-cpu-baseline` on one MI355X (gfx950), BASELINE.json configs[1] "
          "(bg 512^3 + 4 obj 128^3, 640x480).  All 90 frames (30 warm-up + 60 timed) are in the "
          "trace; frame 0 has no raycast/E-step.", ""]
con = db("trace")
stats = {}
if con:
    # bench.py runs the frames twice: measured, then -- behind its copy-bandwidth probe (k_stream_copy) -- once
    # more with the march counters on.  The table is over the measured run only.
    cut = con.execute("select min(start) from kernels where name like '%k_stream_copy%'").fetchone()[0]
    where = f"where start < {cut}" if cut else ""
    rows = con.execute(f"select name, count(*), avg(end-start), min(end-start), max(end-start), "
                       f"sum(end-start) from kernels {where} group by name order by 6 desc").fetchall()
    if cut:
        out_md += ["(launches before the copy-bandwidth probe: the warm-up and the timed frames; the replay with the "
                   "march counters that follows it is left out)", ""]
    total = sum(r[5] for r in rows)
    out_md += ["## Kernel trace (all launches of the run)", "",
               "| kernel | launches | avg us | min us | max us | total ms | share |",
               "|---|---:|---:|---:|---:|---:|---:|"]
    for name, n, avg, mn, mx, tot in rows:
        out_md.append(f"| `{short(name)}` | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
                      f"{tot / 1e6:.2f} | {100 * tot / total:.1f}% |")
        stats[short(name)] = dict(kernel=name.split("(")[0], launches=n, avg_us=avg / 1e3,
                                  total_ms=tot / 1e6)
    out_md.append("")

traffic = defaultdict(dict)
for pas in ("pmc_rd", "pmc_wr", "pmc_fetch", "pmc_write"):
    con = db(pas)
    if not con:
        continue
    rows = con.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) "
                       "from counters_collection group by kernel_name, counter_name").fetchall()
    for kn, cn, v, nd in rows:
        traffic[short(kn)][cn] = (v, nd)
if traffic:
    out_md += ["## HBM-side traffic per launch (TCC_EA0 request counters, separate PMC passes)", "",
               "Read bytes = 32 B x RDREQ_32B + 64 B x RDREQ_64B + 128 B x RDREQ_128B (sized request "
               "counters, so no width assumption is needed); FETCH_SIZE / WRITE_SIZE (KiB) are listed "
               "as reported -- on gfx950 FETCH_SIZE tallies 128-B requests at 64 B "
               "(MI355X_MICROARCH.md, HBM section), which the sized counters make visible.  "
               "Write bytes = 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B).", "",
               "| kernel | launches | read MB/launch | FETCH_SIZE MB/launch | write MB/launch | "
               "WRITE_SIZE MB/launch | L2 hit rate |", "|---|---:|---:|---:|---:|---:|---:|"]
    tj = {}
    for k, c in traffic.items():
        if k in ("memset", "memcpy") or "RDREQ" not in " ".join(c):
            continue
        nd = c.get("TCC_EA0_RDREQ", (0, 1))[1] or 1
        rd = (32 * c.get("TCC_EA0_RDREQ_32B", (0, 1))[0] + 64 * c.get("TCC_EA0_RDREQ_64B", (0, 1))[0] +
              128 * c.get("TCC_EA0_RDREQ_128B", (0, 1))[0]) / nd
        wr_n = c.get("TCC_EA0_WRREQ", (0, 1))
        wr64 = c.get("TCC_EA0_WRREQ_64B", (0, 1))[0]
        wr = (64 * wr64 + 32 * (wr_n[0] - wr64)) / (wr_n[1] or 1)
        fs = c.get("FETCH_SIZE", (0, 1))
        ws = c.get("WRITE_SIZE", (0, 1))
        hit, miss = c.get("TCC_HIT", (0, 1))[0], c.get("TCC_MISS", (0, 1))[0]
        hr = hit / (hit + miss) if hit + miss else float("nan")
        out_md.append(f"| `{k}` | {nd} | {rd / 1e6:.1f} | {fs[0] * 1024 / (fs[1] or 1) / 1e6:.1f} | "
                      f"{wr / 1e6:.1f} | {ws[0] * 1024 / (ws[1] or 1) / 1e6:.1f} | {hr:.2f} |")
        tj[k] = dict(read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                     hbm_bytes_per_launch=rd + wr, l2_hit_rate=hr, launches=nd)
    json.dump(dict(tag=tag, source="rocprofv3 --pmc TCC_EA0_* (scripts/profile_round.sh)",
                   kernels=tj, trace=stats), open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
    out_md.append("")
try:
    line = open(f"{src}/bench_under_trace.json").read().strip()
    if line:
        out_md += ["## bench.py line of the traced run (HIP-event kernel timings inside)", "", "```json",
                   line, "```", ""]
except FileNotFoundError:
    pass
open(f"{dst}/{tag}_rocprof_summary.md", "w").write("\n".join(out_md))
print("\n".join(out_md[:60]))
