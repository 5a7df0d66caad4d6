#!/bin/bash
# like ab.sh, plus the 524288-frame batch (steady state) per variant
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
    python3 $R/bench.py --no-cpu --no-pipelined --steps 30 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 65536', d['roofline']['kernel_ms'], d['value'])"
    python3 $R/bench.py --no-cpu --no-pipelined --steps 10 --warmup 2 --frames 524288 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v 524288', d['roofline']['kernel_ms'], d['value'])"
  done
done
