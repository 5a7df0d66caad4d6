#!/usr/bin/env python3
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
rsdims, nsf = 24, 16384; fb = 192 * rsdims
sym = make_frames(nsf * 5, fb, seed=7, device=dev, ebn0_db=6.0)
d_work = torch.zeros((nsf, 120 * rsdims), dtype=torch.uint8, device=dev)
d_out = torch.zeros((nsf, 110 * rsdims), dtype=torch.uint8, device=dev)
d_ret = torch.zeros(nsf, dtype=torch.int32, device=dev)
def t(fn, steps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / steps
for rep in range(3):
    p = t(lambda: V.dabplus_superframes_dev(sym, d_work, d_out, d_ret, rsdims, nsf))
    d = t(lambda: V.decode_batch_dev(sym, d_work, fb, nsf * 5))
    r = t(lambda: V.rs_batch_dev(d_work, d_out, d_ret, rsdims, nsf))
    print(json.dumps({"pipeline_ms": round(p, 3), "decode_ms": round(d, 3), "rs_ms": round(r, 3), "failed": int((d_ret < 0).sum())}), flush=True)
