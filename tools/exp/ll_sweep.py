#!/usr/bin/env python3
"""needs tools/exp/ll_variant.patch applied (vit_set_low_latency); kept as the script behind profiles/r03_ll_sweep.jsonl"""
"""standard vs low-latency instantiation of the packed kernels over launch sizes (frames per launch -> waves per SIMD)"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize(); V.set_kernel(2)
dev = torch.device("cuda", 0)

def timeit(fn, steps=20):
    t_end = time.perf_counter() + 0.06
    while time.perf_counter() < t_end:
        fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps

for fb, sizes in ((768, (2048, 4096, 6144, 8192, 10240, 12288, 16384, 32768, 65536)), (3072, (1024, 4096, 8192, 10240, 12288, 16384, 32768)),
                  (6912, (1820, 3640, 7280, 10240, 14560, 29120))):
    base = make_frames(max(sizes), fb, seed=fb, device=dev)
    for n in sizes:
        sym = base[:n]
        out = torch.zeros((n, fb // 8), dtype=torch.uint8, device=dev)
        res = {}
        for ll in (0, 1):
            V.set_low_latency(ll)
            res[ll] = timeit(lambda: V.decode_batch_dev(sym, out, fb, n))
        V.set_low_latency(-1)
        print(json.dumps({"framebits": fb, "frames": n, "waves_per_simd": round(n / 4 / 1024, 2), "ms_std": round(res[0], 4), "ms_ll": round(res[1], 4),
                          "Gbit_s_std": round(n * fb / res[0] / 1e6, 1), "Gbit_s_ll": round(n * fb / res[1] / 1e6, 1)}), flush=True)
