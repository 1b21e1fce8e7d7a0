#!/usr/bin/env python
"""Turn the rocprofv3 outputs of scripts/profile_round.sh (rocpd SQLite databases) into the
committed summaries under profiles/: per-kernel launch statistics of the timed bench run, and
per-launch HBM-side traffic of the path's kernels from the TCC_EA0 request counters."""
import glob
import json
import sqlite3
import sys
from collections import OrderedDict, defaultdict

import os as _os
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
EXTRA_ARGS = _os.environ.get("EMF_PROFILE_ARGS", "").strip()
src = f"gpurun_out/prof_{tag}"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles"
SHORT = OrderedDict([("k_raycast_batched", "raycast"), ("k_stream_copy", "stream_copy"), ("k_l1_probe<64>", "l1_probe_64"),
                     ("k_l1_probe<4>", "l1_probe_4"), ("k_l1_probe<1>", "l1_probe_1"), ("k_integrate_listed_rest", "integrate_rest"),
                     ("k_integrate_listed<true>", "integrate_bg"), ("k_integrate_listed<(bool)1>", "integrate_bg"),
                     ("k_integrate_listed", "integrate"), ("k_integrate_cull", "integrate_cull"),
                     ("k_far_bounds_listed", "far_bounds_listed"), ("k_far_bounds", "far_bounds_scan"), ("k_far_init", "far_init"), ("k_sign_maps", "sign_maps"),
                     ("k_relevant_tiles", "relevant_tiles"), ("k_relevant_reset", "relevant_reset"),
                     ("k_integrate_batched", "integrate"),
                     ("k_track_step", "track_step"), ("k_track_prepare", "track_prepare"), ("k_track_maxw", "track_maxw"),
                     ("k_track_weight_images", "track_weight_images"), ("k_pose_gradients", "pose_gradients"),
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
          "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps %s --warmup %s "
          "--no-cpu-baseline%s` (the driver's protocol) on one MI355X (gfx950), %s.  Warm-up and timed frames are in the "
          "trace; frame 0 has no raycast/E-step." % (__import__("os").environ.get("EMF_PROFILE_STEPS", "20"),
                                                     __import__("os").environ.get("EMF_PROFILE_WARMUP", "5"),
                                                     " " + EXTRA_ARGS if EXTRA_ARGS else "",
                                                     "workload arguments `%s`" % EXTRA_ARGS if EXTRA_ARGS else
                                                     "BASELINE.json configs[1] (bg 512^3 + 4 obj 128^3, 640x480)"), ""]
con = db("trace")
stats = {}
if con:
    # bench.py runs the frames twice: measured, then -- behind its copy-bandwidth probe (k_stream_copy) -- once
    # more with the march counters on.  The table is over the measured run only.
    cut = con.execute("select min(start) from kernels where name like '%k_stream_copy%'").fetchone()[0]
    where = f"where start < {cut}" if cut else ""
    import os
    STEPS_, WARMUP_ = int(os.environ.get("EMF_PROFILE_STEPS", "20")), int(os.environ.get("EMF_PROFILE_WARMUP", "5"))
    rows = con.execute(f"select name, count(*), avg(end-start), min(end-start), max(end-start), "
                       f"sum(end-start) from kernels {where} group by name order by 6 desc").fetchall()
    if cut:
        out_md += ["(launches before the copy-bandwidth probe: the warm-up and the timed frames; the replay with the "
                   "march counters that follows it is left out.  `timed avg` = the launches of the timed frames only, "
                   "what bench.py's HIP events average over)", ""]
    total = sum(r[5] for r in rows)
    out_md += ["## Kernel trace (all launches of the run)", "",
               "| kernel | launches | avg us | min us | max us | total ms | share | timed launches | timed avg us |",
               "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    frames = STEPS_ + WARMUP_
    for name, n, avg, mn, mx, tot in rows:
        durs = [r[0] for r in con.execute(f"select end-start from kernels {where + ' and' if where else 'where'} name = ? "
                                          "order by start", (name,)).fetchall()]
        per_frame = 1 if n in (frames, frames - 1) else (round(n / (frames - 1)) if n >= frames - 1 and n % (frames - 1) == 0 else 0)
        timed = durs[-STEPS_ * per_frame:] if per_frame else durs
        tavg = sum(timed) / max(len(timed), 1)
        out_md.append(f"| `{short(name)}` | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
                      f"{tot / 1e6:.2f} | {100 * tot / total:.1f}% | {len(timed)} | {tavg / 1e3:.1f} |")
        stats[short(name)] = dict(kernel=name.split("(")[0], launches=n, avg_us=avg / 1e3,
                                  total_ms=tot / 1e6, timed_launches=len(timed), timed_avg_us=tavg / 1e3)
    out_md.append("")

workload_key = None
try:
    workload_key = json.loads(open(f"{src}/bench_under_trace.json").read().strip())["config"].get("workload_key")
except (OSError, ValueError, KeyError):
    pass

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
    # the tracked workload: bench.py times whole tracking STAGES (one k_track_prepare + the LM launches that follow it), so
    # the stage is the unit its roofline is priced in: per-stage bytes = sum over the tracking kernels / number of stages
    TRACK_KINDS = ("track_step", "track_prepare", "track_maxw", "pose_gradients")
    if "track_prepare" in tj and "track_step" in tj:
        stages = tj["track_prepare"]["launches"]
        agg = {f: sum(tj[k][f] * tj[k]["launches"] for k in TRACK_KINDS if k in tj) / stages
               for f in ("read_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}
        agg.update(l2_hit_rate=tj["track_step"]["l2_hit_rate"], launches=stages,
                   note="per tracking STAGE: sum over " + ", ".join(k for k in TRACK_KINDS if k in tj) + " / stages")
        tj["track"] = agg
    json.dump(dict(tag=tag, workload_key=workload_key, source="rocprofv3 --pmc TCC_EA0_* (scripts/profile_round.sh)",
                   kernels=tj, trace=stats), open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
    out_md.append("")
# ---- issue / L1 / L2 counters of the timed launches (what bench.py's roofline prices the kernels against) ----
import os
STEPS, WARMUP = int(os.environ.get("EMF_PROFILE_STEPS", "20")), int(os.environ.get("EMF_PROFILE_WARMUP", "5"))
counters = defaultdict(dict)
for pas in ("pmc_sq", "pmc_sq2", "pmc_tcp", "pmc_tcc", "pmc_rd", "pmc_wr"):
    con = db(pas)
    if not con:
        continue
    rows = con.execute("select kernel_name, counter_name, dispatch_id, sum(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name, dispatch_id order by dispatch_id").fetchall()
    per = defaultdict(list)
    for kn, cn, did, v, inst in rows:
        per[(short(kn), cn)].append((v, inst))
    for (k, cn), vals in per.items():
        n = len(vals)
        frames = STEPS + WARMUP
        # launches per frame: the timed launches are the last STEPS * per_frame ones (frame 0 has no raycast / E-step)
        per_frame = 1 if n in (frames, frames - 1) else (round(n / (frames - 1)) if n >= frames - 1 and n % (frames - 1) == 0 else 0)
        timed = vals[-STEPS * per_frame:] if per_frame else vals
        counters[k][cn] = dict(per_launch=sum(v for v, _ in timed) / len(timed), instances=timed[0][1],
                               launches=len(timed), of=n)
if "track_prepare" in counters and "track_step" in counters:
    # per tracking stage (see the traffic table): counter totals of the tracking kernels over the number of stages
    KINDS = ("track_step", "track_prepare", "track_maxw", "pose_gradients")
    names = set()
    for k in KINDS:
        names |= set(counters.get(k, {}))
    agg = {}
    for cn in names:
        stages = counters["track_prepare"].get(cn, {}).get("launches")
        if not stages:
            continue
        tot = sum(counters[k][cn]["per_launch"] * counters[k][cn]["launches"] for k in KINDS if cn in counters.get(k, {}))
        agg[cn] = dict(per_launch=tot / stages, instances=counters["track_step"].get(cn, {}).get("instances", 1), launches=stages,
                       of=stages, launches_per_stage={k: counters[k][cn]["launches"] / stages for k in KINDS if cn in counters.get(k, {})})
    counters["track"] = agg
if counters:
    # durations of the same launches from the kernel trace of the PMC passes are perturbed by the counters; the
    # un-perturbed durations are those of the trace pass (`trace` in *_traffic.json) and of bench.py's HIP events
    json.dump(dict(tag=tag, workload_key=workload_key,
                   protocol=f"python bench.py --steps {STEPS} --warmup {WARMUP}{' ' + EXTRA_ARGS if EXTRA_ARGS else ''} (timed launches only)",
                   source="rocprofv3 --pmc, one pass per counter group (scripts/profile_round.sh)",
                   kernels=counters), open(f"{dst}/{tag}_counters.json", "w"), indent=1)
    out_md += ["## Issue / L1 / L2 counters per timed launch", "",
               "| kernel | VALU insts | VALU busy quad-cyc | SALU insts | VMEM rd insts | wave quad-cyc | L1 line accesses | "
               "L1->L2 read req | L2 req | GRBM cycles |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    def g(k, c):
        return counters[k].get(c, {}).get("per_launch", float("nan"))
    for k in counters:
        if k in ("memset", "memcpy"):
            continue
        inst = counters[k].get("GRBM_GUI_ACTIVE", {}).get("instances", 1) or 1
        out_md.append(f"| `{k}` | {g(k, 'SQ_INSTS_VALU'):.3g} | {g(k, 'SQ_ACTIVE_INST_VALU'):.3g} | {g(k, 'SQ_INSTS_SALU'):.3g} | "
                      f"{g(k, 'SQ_INSTS_VMEM_RD'):.3g} | {g(k, 'SQ_WAVE_CYCLES'):.3g} | {g(k, 'TCP_TOTAL_CACHE_ACCESSES_sum'):.3g} | "
                      f"{g(k, 'TCP_TCC_READ_REQ_sum'):.3g} | {g(k, 'TCC_REQ_sum'):.3g} | {g(k, 'GRBM_GUI_ACTIVE') / inst:.3g} |")
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
