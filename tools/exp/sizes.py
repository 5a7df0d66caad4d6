#!/usr/bin/env python3
"""kernel time of a few (framebits, frames) launches for the library in VITERBI_AMD_LIB (A/B of diagnostic builds)"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize(); V.set_kernel(2)
dev = torch.device("cuda", 0)
def timeit(fn, steps=20):
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps
CASES = ((768, 65536), (768, 16384), (3072, 16384), (3072, 81920), (6912, 7280), (6912, 36400))
if os.environ.get("SIZES"):  # e.g. SIZES=3072:81920,6912:36400
    CASES = tuple(tuple(int(x) for x in c.split(":")) for c in os.environ["SIZES"].split(","))
for fb, n in CASES:
    if os.environ.get("SIZES_INPUT") == "random":  # input without signal: the long-frame kernel's waves give up tracing in flight
        sym = torch.randint(0, 256, (n, 4 * (fb + 6)), dtype=torch.uint8, device=dev)
    else:
        sym = make_frames(n, fb, seed=fb, device=dev)
    out = torch.zeros((n, fb // 8), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: V.decode_batch_dev(sym, out, fb, n))
    print(json.dumps({"lib": os.path.basename(os.environ.get("VITERBI_AMD_LIB", "base")), "framebits": fb, "frames": n, "ms": round(ms, 4), "Gbit_s": round(n * fb / ms / 1e6, 1)}), flush=True)
    del sym, out
