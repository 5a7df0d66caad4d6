#!/bin/bash
# Runs on the GPU box (via gpurun): the judged evidence for one round.
#   1. bench.py default run (JSON line)            -> bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> kernel_stats.csv
#   3. PMC passes (separate runs, as the guide prescribes)  -> pmc_*.csv
# usage: tools/collect_profiles.sh <round-tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
# (--no-sensitivity: that leg launches the same kernel on 0 dB and random-byte input, whose longer launches would blur the average;
#  --no-pipelined: the two-stream leg launches overlapping instances of the same kernel, whose durations would blur the
#  average that `roofline.kernel_ms` - the one-stream timed region - has to agree with)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu --no-pipelined --no-sensitivity > $OUT/kt.log 2>&1; echo "kt rc=$?"
cp $OUT/kt/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $R/bench.py --no-cpu --no-pipelined --no-sensitivity --steps 5 --warmup 1 > $OUT/p1.log 2>&1; echo "pmc FETCH rc=$?"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/bench.py --no-cpu --no-pipelined --no-sensitivity --steps 5 --warmup 1 > $OUT/p2.log 2>&1; echo "pmc WRITE rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/p3 -- python3 $R/bench.py --no-cpu --no-rs --no-pipelined --no-sensitivity --steps 5 --warmup 1 > $OUT/p3.log 2>&1; echo "pmc SQ rc=$?"
for p in p1 p2 p3; do cp $OUT/$p/*/*_counter_collection.csv $OUT/pmc_$p.csv 2>/dev/null; done
rm -rf $OUT/kt $OUT/p1 $OUT/p2 $OUT/p3
ls -la $OUT
# optional fourth pass: LDS / memory-instruction counters for the same kernel
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p4 -- python3 $R/bench.py --no-cpu --no-rs --no-pipelined --no-sensitivity --steps 5 --warmup 1 > $OUT/p4.log 2>&1; echo "pmc LDS rc=$?"
cp $OUT/p4/*/*_counter_collection.csv $OUT/pmc_p4.csv 2>/dev/null; rm -rf $OUT/p4
