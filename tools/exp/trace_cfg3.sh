#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_cfg3; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $R/tools/exp/cfg3_overhead.py device_sort > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
out=sys.argv[1]
rows=[]
for f in glob.glob(f"{out}/kt/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
# last 2 iterations: find last occurrences of desc_hist
idx=[i for i,r in enumerate(rows) if "desc_hist" in r[2]]
i0=idx[-2]
t0=rows[i0][0]
for s,e,n in rows[i0:idx[-1]+1]:
    print("%8.1f us  +%6.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
