#!/bin/bash
# A/B of library variants on the RS kernel: tests/tools/bench_rs.py over the error mixes, 131072 and 16384 superframes
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in base "$@"; do
  if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
  for m in clean light rough le2 le3 le5 mixed; do
    python3 $R/tests/tools/bench_rs.py 24 131072 $m 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['nsf'], d['mode'], d['ms'], d['GB_s'], d['parity_ok'])"
  done
  python3 $R/tests/tools/bench_rs.py 24 16384 mixed 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['nsf'], d['mode'], d['ms'], d['GB_s'], d['parity_ok'])"
done
