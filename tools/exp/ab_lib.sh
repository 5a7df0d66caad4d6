#!/bin/bash
# alternating runs of the headline bench over library variants (tools/exp/libviterbi_<name>.so; "base" = the product library)
# with the environment given in AB_ENV (e.g. AB_ENV="VITERBI_AMD_PK8=1")
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = base ]; then L=$R/viterbi.dll_amd/libviterbi.so; else L=$R/tools/exp/libviterbi_$v.so; fi
    env VITERBI_AMD_LIB=$L $AB_ENV python3 $R/bench.py --no-cpu --no-pipelined --no-rs --no-sensitivity --steps 30 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['roofline']['kernel_ms'], d['value'])"
  done
done
