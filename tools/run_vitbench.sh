#!/bin/bash
# builds the native harness on the box and runs it against the in-tree library: tools/run_vitbench.sh [frames] [loops] [sweep]
set -e
mkdir -p gpurun_out
g++ -O2 -std=c++17 -I include tools/vitbench.cpp -o gpurun_out/vitbench.bin -ldl -lpthread
nproc
./gpurun_out/vitbench.bin viterbi.dll_amd/libviterbi.so "${1:-500}" "${2:-2000}" $3
