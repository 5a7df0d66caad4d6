#!/bin/bash
# HBM traffic of the long-frame kernel per launch, by (framebits:frames), for the library in VITERBI_AMD_LIB (default: the product):
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/exp/sizes.py; HBM bytes = 2*FETCH_SIZE*1024 (gfx950
# correction, MI355X_MICROARCH.md) + WRITE_SIZE*1024; algorithmic bytes = frames * (4*(framebits+6) + framebits/8).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_traffic_long; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for S in ${PMC_SIZES:-4608:81920 6912:36400 6912:7280 3072:81920}; do
export SIZES=$S
rm -rf $OUT/a $OUT/b
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/a -- python3 $R/tools/exp/sizes.py > $OUT/a.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/b -- python3 $R/tools/exp/sizes.py > $OUT/b.log 2>&1
python3 - $OUT $S <<'PY'
import csv, glob, sys, os, json
out, S = sys.argv[1:3]
fb, n = (int(x) for x in S.split(":"))
acc = {}
for d, c in (("a", "FETCH_SIZE"), ("b", "WRITE_SIZE")):
    v = []
    for f in glob.glob(f"{out}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "vit_pk_long_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    acc[c] = sum(v) / max(len(v), 1)
rd, wr = 2.0 * acc["FETCH_SIZE"] * 1024, acc["WRITE_SIZE"] * 1024
alg = n * (4 * (fb + 6) + fb // 8)
print(json.dumps({"lib": os.path.basename(os.environ.get("VITERBI_AMD_LIB", "product")), "framebits": fb, "frames": n, "read_MB": round(rd / 1e6, 1), "write_MB": round(wr / 1e6, 1),
                  "algorithmic_MB": round(alg / 1e6, 1), "traffic_over_algorithmic": round((rd + wr) / alg, 3)}), flush=True)
PY
done
