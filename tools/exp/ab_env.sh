#!/bin/bash
# A/B by environment: alternating runs of the headline bench, "name=VAR=value" pairs; e.g. ab_env.sh pk4=VITERBI_AMD_PK8=0 pk8=VITERBI_AMD_PK8=1
R=${GRAFT_REPO_ROOT:-/root/repo}
EXTRA=${AB_BENCH_ARGS:-}
for i in 1 2 3; do
  for v in "$@"; do
    name=${v%%=*}; kv=${v#*=}
    env "$kv" python3 $R/bench.py --no-cpu --no-pipelined --no-rs --steps 30 --warmup 3 $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['roofline']['kernel_ms'], d['value'])"
  done
done
