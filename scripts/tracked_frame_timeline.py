#!/usr/bin/env python
"""Timeline of ONE frame of `bench.py --track` / `--entry` from a rocprofv3 kernel trace (scripts/quick_trace.sh --track):
the kernels between two raycast launches with start time and duration, runs of k_track_step collapsed, and the gaps of
the critical stream -- how round 6 found the 199-us k_mask_mass and the per-object waits of cleanUpObjs.
Usage: tracked_frame_timeline.py [trace.db] [frame]"""
import glob
import re
import sqlite3
import sys

f = sys.argv[1] if len(sys.argv) > 1 else (glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db"))[0]
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 40
con = sqlite3.connect(f)
rows = con.execute("select name, start, end, stream_id from kernels order by start").fetchall()
ray = [i for i, r in enumerate(rows) if "k_raycast_batched" in r[0]]
a, b = ray[frame], ray[frame + 1]
t0 = rows[a][1]


def short(n):
    m = re.search(r"(k_[a-z_0-9]+|__amd_rocclr_[a-zA-Z]+)", n)
    return m.group(1) if m else n[:30]


out = [(short(r[0]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3]) for r in rows[a:b + 1]]
main = out[0][3]
i, prev_end, gaps = 0, None, 0.0
while i < len(out):
    n = out[i]
    j = i
    if n[0] == "k_track_step":
        while j + 1 < len(out) and out[j + 1][0] == "k_track_step":
            j += 1
    end = out[j][1] + out[j][2]
    gap = ""
    if n[3] == main:
        if prev_end is not None and n[1] - prev_end > 3.0:
            gap = f"   <- gap {n[1] - prev_end:6.1f} us on the critical stream"
            gaps += n[1] - prev_end
        prev_end = end
    label = n[0] + (f" x{j - i + 1}" if j > i else "")
    print(f"{label:34s} stream {n[3]}  start {n[1]:8.1f}  end {end:8.1f}  kernels {sum(o[2] for o in out[i:j + 1]):7.1f} us{gap}")
    i = j + 1
print(f"frame {frame}: {out[-1][1]:.1f} us from raycast to raycast; gaps > 3 us on the critical stream: {gaps:.1f} us")
