#!/bin/bash
# alternating runs of tools/exp/sizes.py (kernel 2 = packed, by framebits:frames in SIZES) over library variants
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset VITERBI_AMD_LIB; else export VITERBI_AMD_LIB=$R/tools/exp/libviterbi_$v.so; fi
    python3 $R/tools/exp/sizes.py 2>/dev/null | python3 -c "
import sys, json
r = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('$v', ' '.join('%d:%d=%.4f' % (d['framebits'], d['frames'], d['ms']) for d in r))"
  done
done
