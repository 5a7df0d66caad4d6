#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points (never the headline value)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize(); V.WakeUpYMM()
V.set_renorm_ge(0)  # the oracle's default comparator (`> 150`, the C decoders); the library's default is the MASM decoders' `>= 150`
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

# ---- concurrent callers of deconvolve(): per-call streams vs the micro-batching ingest stage ----
import threading
def run_threads(nthreads, calls):
    syms = [O.noisy_frames(1, fb, seed=10 + i)[0].astype(np.uint32) for i in range(nthreads)]
    outs = [np.zeros(fb // 8, np.uint8) for _ in range(nthreads)]
    def work(i):
        for _ in range(calls):
            V.deconvolve(fb, syms[i], 0, outs[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
    return nthreads * calls / (time.perf_counter() - t0)
for nt in (1, 8, 32):
    V.set_batch_window_us(0); a = run_threads(nt, 300)
    V.set_batch_window_us(50); b = run_threads(nt, 300)
    V.set_batch_window_us(0)
    print(json.dumps({"path": "deconvolve() from %d threads" % nt, "calls_per_s_unbatched": round(a), "calls_per_s_window_50us": round(b)}))
