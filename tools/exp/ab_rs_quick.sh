#!/bin/bash
# quick A/B of library variants on the RS kernel: ab_rs_quick.sh "<modes>" <variant...>  (base = the product library)
R=${GRAFT_REPO_ROOT:-/root/repo}; MODES=$1; shift
for i in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
  for m in $MODES; do
    python3 $R/tests/tools/bench_rs.py 24 131072 $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['nsf'], d['mode'], d['ms'], d['GB_s'], d['parity_ok'])"
  done
  python3 $R/tests/tools/bench_rs.py 24 16384 clean 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['nsf'], d['mode'], d['ms'], d['GB_s'], d['parity_ok'])"
done
done
