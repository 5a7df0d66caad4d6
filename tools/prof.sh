#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (separate runs).
# usage: tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu --no-pipelined --no-sensitivity "$@" > $OUT/kt.log 2>&1
echo "kt rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pipelined --no-sensitivity "$@" > $OUT/pmc1.log 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pipelined --no-sensitivity "$@" > $OUT/pmc2.log 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pipelined --no-sensitivity "$@" > $OUT/pmc3.log 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-pipelined --no-sensitivity "$@" > $OUT/pmc4.log 2>&1
echo "pmc4 rc=$?"
find $OUT -name "*.csv" | head -40
