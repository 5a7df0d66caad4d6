#!/usr/bin/env python3
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
rsdims, nsf = 24, 16384; fb = 192 * rsdims
sym = make_frames(nsf * 5, fb, seed=7, device=dev, ebn0_db=6.0)
d_work = torch.zeros((nsf, 120 * rsdims), dtype=torch.uint8, device=dev)
d_work2 = torch.zeros((nsf, 120 * rsdims), dtype=torch.uint8, device=dev)
d_out = torch.zeros((nsf, 110 * rsdims), dtype=torch.uint8, device=dev)
d_ret = torch.zeros(nsf, dtype=torch.int32, device=dev)
dec = lambda: V.decode_batch_dev(sym, d_work, fb, nsf * 5)
rs = lambda: V.rs_batch_dev(d_work2, d_out, d_ret, rsdims, nsf)
other = lambda: d_out.add_(1)  # an unrelated small torch kernel
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    dec(); torch.cuda.synchronize()
def seq(fns, reps=8):
    ev = []
    for _ in range(reps):
        for f in fns:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    n = len(fns)
    return [round(sum(ev[i * n + k][0].elapsed_time(ev[i * n + k][1]) for i in range(2, reps)) / (reps - 2), 3) for k in range(n)]
print(json.dumps({"decode_only": seq([dec])}))
print(json.dumps({"decode,rs": seq([dec, rs])}))
print(json.dumps({"decode,torch_add": seq([dec, other])}))
print(json.dumps({"decode_only_again": seq([dec])}))
zero = lambda: d_ret.zero_()
print(json.dumps({"decode,zero_": seq([dec, zero])}))
def gap():
    torch.cuda.synchronize(); time.sleep(0.0003)
print(json.dumps({"decode,host_sync+300us_idle": seq([dec, gap])}))
def syncs():
    torch.cuda.synchronize()
print(json.dumps({"decode,host_sync": seq([dec, syncs])}))
fic = make_frames(65536, 768, seed=1, device=dev); fo = torch.zeros((65536, 96), dtype=torch.uint8, device=dev)
f1 = lambda: V.decode_batch_dev(fic, fo, 768, 65536)
print(json.dumps({"fic_only": seq([f1], reps=30)}))
print(json.dumps({"fic,torch_add": seq([f1, other], reps=30)}))
print(json.dumps({"fic,host_sync": seq([f1, syncs], reps=30)}))

