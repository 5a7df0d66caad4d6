#!/usr/bin/env python3
"""kernel time of 16384 / 32768 / 65536 / 262144 FIC frames for the library in VITERBI_AMD_LIB"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
base = make_frames(65536, 768, seed=1, device=dev)
res = {}
for n in (16384, 32768, 65536, 262144):
    sym = base.repeat((n + 65535) // 65536, 1)[:n].contiguous()
    out = torch.zeros((n, 96), dtype=torch.uint8, device=dev)
    for _ in range(20): V.decode_batch_dev(sym, out, 768, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): V.decode_batch_dev(sym, out, 768, n)
    b.record(); torch.cuda.synchronize()
    res[n] = round(a.elapsed_time(b) / 40, 4)
print(os.path.basename(os.environ.get("VITERBI_AMD_LIB", "base")), json.dumps(res), flush=True)
