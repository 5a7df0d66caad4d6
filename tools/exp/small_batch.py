#!/usr/bin/env python3
"""kernel time of small launches: packed kernel vs wave-per-frame kernel"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
for fb in (768, 3072):
    base = make_frames(4096, fb, seed=1, device=dev)
    for n in (1, 2, 4, 16, 64, 256, 1024, 4096):
        sym = base[:n].contiguous(); out = torch.zeros((n, fb // 8), dtype=torch.uint8, device=dev)
        res = {}
        for k in (2, 1):
            V.set_kernel(k)
            for _ in range(3): V.decode_batch_dev(sym, out, fb, n)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): V.decode_batch_dev(sym, out, fb, n)
            b.record(); torch.cuda.synchronize()
            res[k] = a.elapsed_time(b) / 20 * 1e3
        V.set_kernel(0)
        print(json.dumps({"framebits": fb, "frames": n, "us_packed": round(res[2], 1), "us_wave": round(res[1], 1)}), flush=True)
