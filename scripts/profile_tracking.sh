#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of the tracked bench (bench.py --track) and
# a per-kernel table + the launch sequence of one tracking stage -> gpurun_out/summary/$1_tracking_kernels.md
tag=${1:-r02d}
out=gpurun_out/prof_${tag}_track
rm -rf $out; mkdir -p $out gpurun_out/summary
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 60 --warmup 30 --no-cpu-baseline --no-stats-replay --track"
timeout 300 rocprofv3 --kernel-trace --stats -d $out -o t -- $CMD > $out/trace.log 2>&1; echo "trace rc=$?"
python bench.py --steps 100 --warmup 30 --no-cpu-baseline --track 2>/dev/null | grep '"metric"' > $out/bench_untraced.json
python - "$tag" "$out" "$CMD" <<'PY'
import glob, json, re, sqlite3, sys
tag, out, cmd = sys.argv[1:4]
con = sqlite3.connect((glob.glob(out + "/*.db") + glob.glob(out + "/*/*.db"))[0])
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", n)
    return m.group(1) if m else n[:40]
rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                   "from kernels group by name order by 6 desc").fetchall()
total = sum(r[5] for r in rows)
md = [f"# rocprofv3 kernel trace of the tracked frame, round tag `{tag}`", "",
      f"Command: `rocprofv3 --kernel-trace --stats -- {cmd}` on one MI355X (gfx950): BASELINE.json configs[1] "
      "(bg 512^3 + 4 obj 128^3, 640x480) with the camera and the four objects tracked by the device-resident "
      "LM-ICP (SURVEY 8 f-1) instead of taking supplied poses.  90 frames in the trace.", "",
      "| kernel | launches | avg us | min us | max us | total ms | share |", "|---|---:|---:|---:|---:|---:|---:|"]
for name, n, avg, mn, mx, tot in rows[:22]:
    md.append(f"| `{short(name)}` | {n} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {tot / 1e6:.2f} | {100 * tot / total:.1f}% |")
# one camera stage: the launches between a k_track_prepare and the next one
seq = con.execute("select name, start, end from kernels order by start").fetchall()
prep = [i for i, r in enumerate(seq) if "k_track_prepare" in r[0]]
i0, i1 = prep[len(prep) // 2 // 2 * 2], prep[len(prep) // 2 // 2 * 2 + 1]
steps = [r for r in seq[i0:i1] if "k_track_step" in r[0]]
dur = [(r[2] - r[1]) / 1e3 for r in steps]
busy = [d for d in dur if d > 8]
md += ["", "## One camera stage (launch sequence between two `k_track_prepare`)", "",
       f"{len(steps)} `k_track_step` launches: {len(busy)} that do an LM iteration (median {sorted(busy)[len(busy) // 2]:.1f} us), "
       f"{len(dur) - len(busy)} with little or nothing left to do (median {sorted(d for d in dur if d <= 8)[max(0, (len(dur) - len(busy)) // 2 - 0)] if len(dur) > len(busy) else 0:.1f} us); "
       f"stage span {(steps[-1][2] - seq[i0][1]) / 1e3:.0f} us, of which kernels {sum(dur):.0f} us.", "",
       "durations (us): " + " ".join(f"{d:.0f}" for d in dur), ""]
for f in ("trace.log",):
    for line in open(out + "/" + f):
        if '"metric"' in line:
            md += ["## bench.py line of the traced run", "", "```json", line.strip(), "```", ""]
try:
    md += ["## bench.py line of an untraced run right after (100 steps)", "", "```json", open(out + "/bench_untraced.json").read().strip(), "```", ""]
except FileNotFoundError:
    pass
open(f"gpurun_out/summary/{tag}_tracking_kernels.md", "w").write("\n".join(md))
print("\n".join(md[:40]))
PY
rm -rf $out/*.db $out/*/*.db
