#!/bin/bash
# per-wave SQ counters of the kernels matching <pattern> while running a command: pmc_cmd.sh <tag> <pattern> <python-script> [args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PAT=$2; shift 2
OUT=$R/gpurun_out/pmc_cmd; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/$TAG -- python3 "$@" > $OUT/$TAG.log 2>&1
python3 - $OUT/$TAG "$PAT" $TAG <<'PY'
import csv, glob, sys, collections
d, pat, tag = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{d}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:60], r["Grid_Size"])
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key, c in acc.items():
    waves = c["SQ_WAVES"][0] / max(c["SQ_WAVES"][1], 1)
    print(tag, key, "launches", c["SQ_WAVES"][1], "waves", waves, {k: round(a[0] / a[1] / waves, 1) for k, a in sorted(c.items()) if k != "SQ_WAVES"})
PY
