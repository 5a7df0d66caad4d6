#!/bin/bash
# Runs on the GPU box: secondary evidence (configs 3/4/5, RS, host paths, native harness, soak) + a
# rocprofv3 kernel-trace of the config run so the long-frame, sort and RS kernels have measured durations.
# usage: tools/collect_config_profiles.sh <round-tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/configs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tests/tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err; echo "configs rc=$?"
for m in clean light rough le2 le5 mixed; do python3 $R/tests/tools/bench_rs.py 24 131072 $m 2>/dev/null; python3 $R/tests/tools/bench_rs.py 24 16384 $m 2>/dev/null; done > $OUT/rs.jsonl; echo "rs rc=$?"
python3 $R/tools/exp/small_batch.py > $OUT/small_batch.jsonl 2>/dev/null; echo "small_batch rc=$?"
python3 $R/tests/tools/bench_host_paths.py > $OUT/hostpaths.jsonl 2>/dev/null; echo "host rc=$?"
[ -x $R/tools/vitbench.bin ] || g++ -O2 -std=c++17 -I $R/include -o $R/tools/vitbench.bin $R/tools/vitbench.cpp -ldl -lpthread
$R/tools/vitbench.bin $R/viterbi.dll_amd/libviterbi.so > $OUT/vitbench.txt 2>&1; echo "vitbench rc=$?"
python3 $R/tests/tools/soak.py 5 > $OUT/soak.jsonl 2>/dev/null; echo "soak rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tests/tools/bench_configs.py > $OUT/kt.log 2>&1; echo "kt rc=$?"
cp $OUT/kt/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $R/tests/tools/bench_configs.py > $OUT/p1.log 2>&1; echo "pmc FETCH rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python3 $R/tests/tools/bench_configs.py > $OUT/p2.log 2>&1; echo "pmc WRITE rc=$?"
for p in p1 p2; do cp $OUT/$p/*/*_counter_collection.csv $OUT/pmc_$p.csv 2>/dev/null; done
rm -rf $OUT/kt $OUT/p1 $OUT/p2
ls -la $OUT
