#!/usr/bin/env python3
"""kernel time vs batch size for FIC frames (fixed overhead / tail-effect study)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
base = make_frames(65536, 768, seed=1, device=dev)
for n in (4096, 8192, 16384, 32768, 49152, 61440, 65536, 69632, 81920, 131072, 262144, 524288):
    sym = base.repeat((n + 65535) // 65536, 1)[:n].contiguous()
    out = torch.zeros((n, 96), dtype=torch.uint8, device=dev)
    for _ in range(3): V.decode_batch_dev(sym, out, 768, n)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): V.decode_batch_dev(sym, out, 768, n)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(json.dumps({"frames": n, "ms": round(ms, 4), "Mbit_s": round(n * 768 / ms / 1e3, 1), "us_per_4096_waves": round(ms * 1e3 / (n / 16384), 1)}), flush=True)
