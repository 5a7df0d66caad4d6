#!/usr/bin/env python3
"""One-off soak: many frames, GPU (auto kernel) vs the CPU oracle, bit-exact.  Run on the GPU box."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize()
dev = torch.device("cuda", 0)
ncpu = len(os.sched_getaffinity(0))
total = bad = 0
t0 = time.time()
for fb, n in ((768, 262144), (288, 131072), (1536, 65536), (3072, 32768), (6912, 16384), (9216, 8192), (776, 32768), (784, 32768)):
    for kind in ("noisy3dB", "noisy0dB", "uniform"):
        if kind == "uniform":
            sym = torch.randint(0, 256, (n, 4 * (fb + 6)), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(fb))
        else:
            sym = make_frames(n, fb, seed=fb + len(kind), device=dev, ebn0_db=3.0 if kind == "noisy3dB" else 0.0)
        out = torch.zeros((n, fb // 8), dtype=torch.uint8, device=dev)
        V.decode_batch_dev(sym, out, fb, n); torch.cuda.synchronize()
        ref = O.decode_batch(fb, sym.cpu().numpy(), nthreads=ncpu, avx2=O.has_avx2())
        nb = int((out.cpu().numpy() != ref).any(axis=1).sum())
        total += n; bad += nb
        print(json.dumps({"framebits": fb, "kind": kind, "frames": n, "differing": nb}), flush=True)
print(json.dumps({"total_frames": total, "differing": bad, "seconds": round(time.time() - t0, 1)}))
