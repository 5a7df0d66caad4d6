#!/usr/bin/env python3
"""kernel time of small launches: packed kernel vs latency kernel vs wave-per-frame kernel (back-to-back launches,
so the GPU is at its working clocks: a lone deconvolve() call between idle gaps runs at lower clocks)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
for fb in (768, 3072):
    base = make_frames(16384, fb, seed=1, device=dev)
    for n in (1, 4, 64, 256, 1024, 2048, 4096, 8192, 16384):
        assert n <= base.shape[0]
        sym = base[:n].contiguous(); out = torch.zeros((n, fb // 8), dtype=torch.uint8, device=dev)
        res = {}
        for k in (2, 3, 1):
            V.set_kernel(k)
            for _ in range(3): V.decode_batch_dev(sym, out, fb, n)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50): V.decode_batch_dev(sym, out, fb, n)
            b.record(); torch.cuda.synchronize()
            res[k] = a.elapsed_time(b) / 50 * 1e3
        V.set_kernel(0)
        print(json.dumps({"framebits": fb, "frames": n, "us_packed": round(res[2], 1), "us_latency": round(res[3], 1), "us_wave": round(res[1], 1)}), flush=True)
