#!/bin/bash
# per-wave instruction and LDS counters of rs_kernel for one error mix, per library variant: rs_pmc.sh <mode> <variant...>
R=${GRAFT_REPO_ROOT:-/root/repo}; MODE=$1; shift
OUT=$R/gpurun_out/rs_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/a_$v -- python3 $R/tests/tools/bench_rs.py 24 131072 $MODE > $OUT/a_$v.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b_$v -- python3 $R/tests/tools/bench_rs.py 24 131072 $MODE > $OUT/b_$v.log 2>&1
  python3 - $OUT $v $MODE <<'PY'
import csv, glob, sys, collections
out, v, mode = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{out}/a_{v}/*/*counter_collection.csv") + glob.glob(f"{out}/b_{v}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rs_kernel" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
waves = acc["SQ_WAVES"][0] / max(acc["SQ_WAVES"][1], 1)
print(v, mode, "launches", acc["SQ_WAVES"][1], "waves", waves, " per wave:", {k: round(a[0] / a[1] / waves, 1) for k, a in sorted(acc.items()) if k != "SQ_WAVES"})
PY
done
