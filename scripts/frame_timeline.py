"""Per-kernel timeline of one bench frame from the rocprofv3 database scripts/quick_trace.sh leaves in
gpurun_out/quick_trace/: start (us after the frame's raycast), duration, queue, idle gap in front of it."""
import glob, re, sqlite3, sys
f = sys.argv[1:] or glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db")
con = sqlite3.connect(f[0])
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
ray = [i for i, r in enumerate(rows) if "k_raycast_batched" in r[0]]
i0, i1 = ray[len(ray) // 2], ray[len(ray) // 2 + 1]
t0 = rows[i0][1]
print("frame period us", (rows[i1][1] - t0) / 1e3)
prev = {}
for r in rows[i0 - 14:i1 + 1]:
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", r[0])
    q = r[3]
    gap = (r[1] - prev.get(q, r[1])) / 1e3
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}us  q{q} gap {gap:6.1f}  {m.group(1) if m else r[0][:30]}")
    prev[q] = r[2]
