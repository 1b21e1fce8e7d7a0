#!/bin/bash
# Usage (GPU box): quick_trace.sh [bench args]  -- rocprofv3 kernel trace of a short bench run, per-kernel table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/quick_trace
rm -rf $out; mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --stats -d $out -o t -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline --no-stats-replay "$@" > $out/log.txt 2>&1
python - <<'PY'
import glob, sqlite3
f = glob.glob("gpurun_out/quick_trace/*.db") + glob.glob("gpurun_out/quick_trace/*/*.db")
con = sqlite3.connect(f[0])
rows = con.execute("select name, count(*), avg(end-start), max(end-start), sum(end-start) from kernels group by name order by 5 desc").fetchall()
for name, n, avg, mx, tot in rows[:24]:
    import re
    mm = re.search(r"(k_\w+(<[^>]*>)?|__amd_\w+)", name)
    print(f"{(mm.group(1) if mm else name[:50]):44s} n={n:5d} avg={avg/1e3:8.1f}us max={mx/1e3:8.1f}us total={tot/1e6:8.2f}ms")
PY
grep '"metric"' $out/log.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'],'fps',d['ms_per_step'],'ms')"
