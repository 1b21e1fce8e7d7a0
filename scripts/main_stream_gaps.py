import sqlite3, re, glob
import numpy as np
f = glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db")
con = sqlite3.connect(f[0])
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
def short(n):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", n); return m.group(1) if m else n[:30]
ray=[i for i,r in enumerate(rows) if 'k_raycast_batched' in r[0]]
out=[]
for k in range(len(ray)//2, len(ray)//2+20):
    i0,i1=ray[k],ray[k+1]
    seq=[(short(r[0]),r[1],r[2],r[3]) for r in rows[i0:i1+1]]
    main=[s for s in seq if s[3]==seq[0][3]]
    out.append([(main[j+1][0], (main[j+1][1]-main[j][2])/1e3) for j in range(len(main)-1)])
names=[n for n,_ in out[0]]
print('main-stream gaps (median us):', [(n, round(float(np.median([g[j][1] for g in out if len(g)==len(names)])),1)) for j,n in enumerate(names)])
i0,i1=ray[len(ray)//2],ray[len(ray)//2+1]
t0=rows[i0][1]
for r in rows[i0:i1+1]:
    print(f"{(r[1]-t0)/1e3:8.1f} {(r[2]-r[1])/1e3:8.1f} q{r[3]} {short(r[0])}")
