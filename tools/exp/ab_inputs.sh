#!/bin/bash
# alternating runs of tests/tools/bench_inputs.py (configs 2 and 3 x four input families) over library variants:
#   ab_inputs.sh <variant...>   (base = the product library; tools/exp/libviterbi_<name>.so otherwise)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
    python3 $R/tests/tools/bench_inputs.py 2>/dev/null | python3 -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$v', ' '.join('%.4f' % d['ms'] for d in r), 'ok' if all(d['parity_sample_ok'] for d in r) else 'PARITY')"
  done
done
