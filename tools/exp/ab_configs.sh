#!/bin/bash
# alternating runs of tests/tools/bench_configs.py (decode cases only) over library variants
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
    python3 $R/tests/tools/bench_configs.py 2>/dev/null | python3 -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
r = [d for d in r if 'ms' in d and ('case' in d) and not d['case'].startswith('config5 RS')]
print('$v', ' | '.join('%s=%.4f' % (d['case'][:14] + str(d.get('framebits', '')), d['ms']) for d in r))"
  done
done
