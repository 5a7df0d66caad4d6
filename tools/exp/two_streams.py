#!/usr/bin/env python3
"""the 65536-frame FIC batch decoded K times: back to back on ONE stream vs alternating on TWO streams (two output buffers):
does the fixed ~36 us of a launch (ramp + tail) disappear when the next launch fills the chip while the last one drains?"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
n = 65536
sym = make_frames(n, 768, seed=1, device=dev)
outs = [torch.zeros((n, 96), dtype=torch.uint8, device=dev) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
def run(nstreams, K=300):
    for s in streams: s.synchronize()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for i in range(8): V.decode_batch_dev(sym, outs[0], 768, n)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        s = streams[k % nstreams]
        V.decode_batch_dev(sym, outs[k % nstreams], 768, n, stream=s.cuda_stream)
    for s in streams: s.synchronize()
    dt = time.perf_counter() - t0
    return dt / K * 1e3
for ns in (1, 2, 3, 1, 2, 3):
    ms = run(ns)
    print(json.dumps({"streams": ns, "ms_per_batch": round(ms, 4), "Gbit_s": round(n * 768 / ms / 1e6, 1)}), flush=True)
print(json.dumps({"all_equal": bool(torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))}))
