"""k_track_step launches of a traced tracked bench (the database `scripts/quick_trace.sh --track` leaves): per stage
class (camera = one model, objects = the rest) launches per frame, their mean duration, how many of them were
idle (a model-less launch behind the stage's end, < 6 us), the stage span per frame."""
import glob, sqlite3, sys
f = sys.argv[1:] or glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db")
con = sqlite3.connect(f[0])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
gy = "grid_size_y" if "grid_size_y" in cols else ("grid_y" if "grid_y" in cols else None)
wy = "workgroup_size_y" if "workgroup_size_y" in cols else None
rows = con.execute("select name, start, end%s from kernels order by start" % (", " + gy if gy else "")).fetchall()
nray = sum(1 for r in rows if "k_raycast_batched" in r[0])
stages, cur = [], None
for r in rows:
    if "k_track_step" in r[0]:
        if cur is None: cur = dict(t0=r[1], n=0, idle=0, busy=0.0, gy=r[3] if gy else 0)
        cur["n"] += 1; cur["t1"] = r[2]; d = (r[2] - r[1]) / 1e3
        if d < 6: cur["idle"] += 1
        else: cur["busy"] += d
    elif cur is not None and "k_track" not in r[0]:
        stages.append(cur); cur = None
if cur: stages.append(cur)
print("frames (raycasts):", nray, " stages:", len(stages), " grid-y column:", gy)
for cls in sorted(set(s["gy"] for s in stages)):
    ss = [s for s in stages if s["gy"] == cls]
    n = sum(s["n"] for s in ss); idle = sum(s["idle"] for s in ss); busy = sum(s["busy"] for s in ss)
    span = sum((s["t1"] - s["t0"]) / 1e3 for s in ss)
    print("grid y %s: %d stages, %.1f launches per stage (%.1f idle), working launch %.2f us, span %.1f us per stage" % (
        cls, len(ss), n / len(ss), idle / len(ss), busy / max(1, n - idle), span / len(ss)))
