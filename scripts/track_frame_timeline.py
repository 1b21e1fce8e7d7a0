"""Timeline of one TRACKED bench frame from the rocprofv3 database `scripts/quick_trace.sh --track` leaves in
gpurun_out/quick_trace/: every kernel that is not a k_track_step launch (start in us after the frame's raycast,
duration, queue, idle gap in front of it on its queue), the tracking stages as one line each, and the idle time of
the device between the kernels of the frame (what the host round trips at the stages' ends cost)."""
import glob, re, sqlite3, sys
f = sys.argv[1:] or glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db")
con = sqlite3.connect(f[0])
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
ray = [i for i, r in enumerate(rows) if "k_raycast_batched" in r[0]]
i0, i1 = ray[len(ray) // 2], ray[len(ray) // 2 + 1]
t0 = rows[i0][1]
print("frame period us %.1f" % ((rows[i1][1] - t0) / 1e3))
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", n)
    return m.group(1) if m else n[:30]
busy_end, idle, stage = t0, 0.0, None
for r in rows[i0:i1 + 1]:
    name = short(r[0])
    if r[1] > busy_end:
        gap = (r[1] - busy_end) / 1e3
        idle += gap
        if gap > 8:
            print(f"{(busy_end - t0) / 1e3:9.1f}  -- device idle for {gap:.1f} us --")
    busy_end = max(busy_end, r[2])
    if "k_track_step" in name:
        if stage is None:
            stage = [r[1], r[2], 1, r[2] - r[1]]
        else:
            stage[1], stage[2], stage[3] = r[2], stage[2] + 1, stage[3] + r[2] - r[1]
        continue
    if stage is not None:
        print(f"{(stage[0] - t0) / 1e3:9.1f} {(stage[1] - stage[0]) / 1e3:8.1f}us  {stage[2]} x k_track_step, kernels {stage[3] / 1e3:.1f} us")
        stage = None
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}us  q{r[3]}  {name}")
print("device idle inside the frame: %.1f us" % idle)
