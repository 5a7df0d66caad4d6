#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (never the headline value)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import _vitpkg
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize(); V.WakeUpYMM()
fb = 768
sym = O.noisy_frames(1, fb, seed=1)[0].astype(np.uint32)
out = np.zeros(fb // 8, np.uint8)
for _ in range(50): V.deconvolve(fb, sym, 0, out)
t0 = time.perf_counter(); n = 2000
for _ in range(n): V.deconvolve(fb, sym, 0, out)
dt = (time.perf_counter() - t0) / n
print(json.dumps({"path": "deconvolve() single frame, host u32 buffers", "us_per_call": round(dt * 1e6, 1), "Mbit_s": round(fb / dt / 1e6, 2)}))
nf = 65536
syms = np.tile(O.noisy_frames(256, fb, seed=2), (nf // 256, 1))
V.decode_batch_host(syms, fb)
t0 = time.perf_counter()
for _ in range(3): V.decode_batch_host(syms, fb)
dt = (time.perf_counter() - t0) / 3
print(json.dumps({"path": "vit_decode_batch_host 65536 FIC frames (pageable host memory, H2D+kernel+D2H)", "ms": round(dt * 1e3, 2), "Mbit_s": round(nf * fb / dt / 1e6, 1), "GB_s_in": round(syms.nbytes / dt / 1e9, 2)}))
