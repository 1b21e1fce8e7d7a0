"""Durations (us) of the n-th launch of the frame's two long kernels from the rocprofv3 database scripts/quick_trace.sh
leaves in gpurun_out/quick_trace/: how the raycast and the background's integration develop over the first frames."""
import glob, sqlite3
f = glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db")
con = sqlite3.connect(f[0])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
ray = [(e - s) / 1e3 for n, s, e in rows if "k_raycast_batched" in n]
integ = [(e - s) / 1e3 for n, s, e in rows if "k_integrate_listed<" in n]
cull = [(e - s) / 1e3 for n, s, e in rows if "k_integrate_cull" in n]
print("launch raycast integrate_bg cull")
for i in range(0, min(len(ray), 70), 2):
    print("%3d %7.1f %7.1f %6.1f" % (i + 1, ray[i], integ[i + 1] if i + 1 < len(integ) else 0, cull[i + 1] if i + 1 < len(cull) else 0))
