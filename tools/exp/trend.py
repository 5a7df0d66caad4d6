#!/usr/bin/env python3
"""per-launch kernel time of the headline batch over the first 60 launches (warm-up trend)"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
sym = make_frames(65536, 768, seed=1234, device=dev)
out = torch.zeros((65536, 96), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
for a, b in evs:
    a.record(); V.decode_batch_dev(sym, out, 768, 65536); b.record()
torch.cuda.synchronize()
t = [round(a.elapsed_time(b) * 1e3, 1) for a, b in evs]
print(json.dumps({"us": t}))
import time
time.sleep(2.0)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for a, b in evs:
    a.record(); V.decode_batch_dev(sym, out, 768, 65536); b.record()
torch.cuda.synchronize()
print(json.dumps({"after_2s_idle_us": [round(a.elapsed_time(b) * 1e3, 1) for a, b in evs]}))
