#!/bin/bash
# Usage (on the GPU box via gpurun): pmc_kernels.sh TAG "COUNTER ..." ["COUNTER ..." ...]
# One rocprofv3 --pmc pass per counter group over a short bench run; prints per-kernel averages.
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PMCBENCH="python bench.py --steps 10 --warmup 30 --no-cpu-baseline --no-kernel-events"
i=0
for group in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $group --kernel-trace -d $out/g$i -o p -- $PMCBENCH > $out/g$i.log 2>&1; echo "group $i ($group) rc=$?"
  python - "$out/g$i" <<'PY'
import sqlite3, sys, glob
for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    rows = con.execute("""
        SELECT kernel_name, counter_name, SUM(value), COUNT(DISTINCT dispatch_id)
        FROM counters_collection GROUP BY kernel_name, counter_name""").fetchall()
    import re
    for name, ctr, total, n in rows:
        m = re.search(r"k_\w+", name)
        short = m.group(0) if m else name[:40]
        if any(s in short for s in ("integrate_batched", "raycast_batched", "estep")):
            print(f"  {short:40s} {ctr:28s} per-launch {total / max(n,1):16.1f}  launches {n}")
PY
  rm -rf $out/g$i
done
