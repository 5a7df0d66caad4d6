#!/bin/bash
# SQ counters of the packed kernels by launch (tools/exp/sizes.py: FIC and long frames, one round and five): VALU / LDS / wait profile of the long-frame kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_long; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for S in ${PMC_SIZES:-768:65536 3072:16384 3072:81920 6912:36400}; do
export SIZES=$S
rm -rf $OUT/a $OUT/b
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $R/tools/exp/sizes.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 $R/tools/exp/sizes.py > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{out}/a/*/*counter_collection.csv") + glob.glob(f"{out}/b/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "vit_pk" not in r["Kernel_Name"]: continue
        key = ("long" if "vit_pk_long_kernel" in r["Kernel_Name"] else "short", r["Grid_Size"])
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key, c in sorted(acc.items()):
    m = {k: a[0] / a[1] for k, a in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print(__import__("os").environ.get("SIZES"), key, "launches", c["SQ_WAVES"][1], "waves", m.get("SQ_WAVES"), "cycles %.0f" % cyc,
          "VALU busy %.3f" % (m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc) if cyc else 0),
          "LDS inst busy %.3f" % (m.get("SQ_ACTIVE_INST_LDS", 0) * 4 / (256 * cyc) if cyc else 0),
          "LDS idx busy %.3f" % (m.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc) if cyc else 0),
          "VALU insts %.3e" % m.get("SQ_INSTS_VALU", 0), "VMEM insts %.3e" % m.get("SQ_INSTS_VMEM", 0), "SALU %.3e" % m.get("SQ_INSTS_SALU", 0))
PY
done
