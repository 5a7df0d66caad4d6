#!/usr/bin/env python3
"""How much the decode rate depends on the INPUT: BASELINE config 2 (65536 FIC frames) and config 3
(32768 mixed lengths 288..6912) on reference-style noise (Eb/N0 = 3 dB), on a poor channel (0 dB), on uniform random bytes
and on random hard decisions - the last two are a channel without any signal: the survivor paths merge late and the
speculative traceback has to work harder (profiles/r03_merge_depth.txt).  One JSON object per line; parity on a sample."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg  # noqa: E402
from bench import make_frames  # noqa: E402

V = _vitpkg.load_package()
V.set_renorm_ge(0)  # the oracle's default comparator (`> 150`, the C decoders); the library's default is the MASM decoders' `>= 150`
O = _vitpkg.load_oracle()
dev = torch.device("cuda", 0)
V.initialize()


def timeit(fn, steps=20, prewarm_ms=150.0):
    t_end = time.perf_counter() + prewarm_ms / 1e3
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def family(label, n, fb, seed):
    if label.startswith("noisy"):
        return make_frames(n, fb, seed=seed, device=dev, ebn0_db=3.0 if "3 dB" in label else 0.0)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if label == "random bytes":
        return torch.randint(0, 256, (n, 4 * (fb + 6)), generator=g, dtype=torch.uint8, device=dev)
    return (torch.randint(0, 2, (n, 4 * (fb + 6)), generator=g, dtype=torch.uint8, device=dev) * 255).to(torch.uint8)


LABELS = ("noisy Eb/N0 3 dB", "noisy Eb/N0 0 dB", "random bytes", "random hard 0/255")
# ---- config 2 ----
n, fb = 65536, 768
ref = None
for label in LABELS:
    sym = family(label, n, fb, 11)
    out = torch.zeros((n, fb // 8), dtype=torch.uint8, device=dev)
    ms = timeit(lambda: V.decode_batch_dev(sym, out, fb, n))
    ok = bool(np.array_equal(out[:256].cpu().numpy(), O.decode_batch(fb, sym[:256].cpu().numpy(), nthreads=16)))
    ref = ref or ms
    print(json.dumps({"case": "config2: 65536 FIC frames", "symbols": label, "ms": round(ms, 4), "Gbit_s": round(n * fb / ms / 1e6, 1),
                      "vs_3dB": round(ref / ms, 3), "parity_sample_ok": ok}), flush=True)
# ---- config 3 ----
rng = np.random.default_rng(3)
n = 32768
fbs = 96 * rng.integers(3, 73, n)
desc, sym_bytes, out_bytes = V.make_descs(fbs.tolist())
d_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
so_all = torch.from_numpy(desc["sym_offset"].astype(np.int64)).to(dev)
mx = int(os.environ.get("CFG3_MAX", fbs.max()))  # CFG3_MAX: a caller that declares a generous max_framebits (e.g. 9216)
ref = None
for label in LABELS:
    sym = torch.empty(sym_bytes, dtype=torch.uint8, device=dev)
    for m in range(3, 73):
        idx = torch.from_numpy(np.nonzero(fbs == 96 * m)[0]).to(dev)
        if idx.numel():
            fr = family(label, int(idx.numel()), 96 * m, 300 + m)
            pos = so_all[idx][:, None] + torch.arange(fr.shape[1], device=dev)[None, :]
            sym[pos.reshape(-1)] = fr.reshape(-1)
            del fr, pos
    out = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
    ms = timeit(lambda: V.decode_varlen_dev(sym, out, d_desc, n, mx), steps=10, prewarm_ms=200.0)
    sh, oh = sym.cpu().numpy(), out.cpu().numpy()
    ok = True
    for i in rng.choice(n, 48, replace=False):
        f = int(fbs[i]); so = int(desc["sym_offset"][i]); oo = int(desc["out_offset"][i])
        ok &= bool(np.array_equal(O.decode_batch(f, sh[so:so + O.sym_len(f)])[0], oh[oo:oo + f // 8]))
    ref = ref or ms
    print(json.dumps({"case": "config3: 32768 frames of 288..6912 bits", "symbols": label, "ms": round(ms, 3),
                      "Gbit_s": round(float(fbs.sum()) / ms / 1e6, 1), "vs_3dB": round(ref / ms, 3), "parity_sample_ok": ok}), flush=True)
    del sym
