#!/bin/bash
# per-wave instruction / issue counters of the packed decode kernel of the headline bench, per configuration:
#   pk_pmc.sh name=ENVVAR=value ...   (e.g. pk4=VITERBI_AMD_PK8=0 pk8=VITERBI_AMD_PK8=1; VITERBI_AMD_LIB=... works too)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pk_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%=*}; kv=${v#*=}
  export "$kv"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/a_$name -- python3 $R/bench.py --no-cpu --no-pipelined --no-rs --no-sensitivity --steps 3 --warmup 1 $PK_PMC_ARGS > $OUT/a_$name.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/b_$name -- python3 $R/bench.py --no-cpu --no-pipelined --no-rs --no-sensitivity --steps 3 --warmup 1 $PK_PMC_ARGS > $OUT/b_$name.log 2>&1
  unset "${kv%%=*}"
  python3 - $OUT $name <<'PY'
import csv, glob, sys, collections
out, v = sys.argv[1:3]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{out}/a_{v}/*/*counter_collection.csv") + glob.glob(f"{out}/b_{v}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vit_pk" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
waves = acc["SQ_WAVES"][0] / max(acc["SQ_WAVES"][1], 1)
per = {k: round(a[0] / a[1] / waves, 1) for k, a in sorted(acc.items()) if k != "SQ_WAVES"}
print(v, "launches", acc["SQ_WAVES"][1], "waves", waves, " per wave:", per)
if "SQ_ACTIVE_INST_VALU" in per and "GRBM_GUI_ACTIVE" in acc:
    cyc = acc["GRBM_GUI_ACTIVE"][0] / acc["GRBM_GUI_ACTIVE"][1]
    print(v, "  kernel cycles %.0f; VALU busy = SQ_ACTIVE_INST_VALU*4/(1024 SIMD*cycles) = %.3f" % (cyc, per["SQ_ACTIVE_INST_VALU"] * waves * 4 / (1024 * cyc)))
PY
done
