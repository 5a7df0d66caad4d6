#!/usr/bin/env python3
"""Soak of the long-frame kernel's in-flight parts (TEST INFRASTRUCTURE: the CPU oracle is the checker).

Descriptor tables and uniform batches of DAB-sized frames carrying reference-style noise at Eb/N0 = 4 ... 1 dB, i.e. the regime in
which waves keep tracing in flight and 1 ... 30 % of their parts fail the check and come back from the spill (soak_varlen.py's
uniform bytes make every wave give up after its first part).  Per table a pool of frames per (length, Eb/N0) is decoded by the
oracle and tiled; every output byte of the launch is compared with the pool frame it is a copy of.
usage: python tests/tools/soak_inflight.py [tables]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); O.build(); V.initialize()
V.set_renorm_ge(0)
ntab = int(sys.argv[1]) if len(sys.argv) > 1 else 8
total = bad = 0
t0 = time.time()
for seed in range(ntab):
    rng = np.random.default_rng(7000 + seed)
    nlen = int(rng.integers(2, 9))
    lens = sorted(set((96 * rng.integers(9, 97, nlen)).tolist()) | ({784, 800} if seed % 3 == 0 else set()))
    pool = {}
    for fb in lens:
        for k, db in enumerate((4.0, 3.0, 2.5, 2.0, 1.5, 1.0)):
            s = O.noisy_frames(6, fb, seed=seed * 1000 + fb + k, ebn0_db=db)
            pool[(fb, k)] = (s, O.decode_batch(fb, s, nthreads=8))
    n = int(rng.choice([600, 4096, 16384, 30000]))
    uniform = seed % 4 == 1
    if uniform:  # one length, the batch entry point
        fb = lens[-1]
        pick = [(fb, int(k), int(j)) for k, j in zip(rng.integers(0, 6, n), rng.integers(0, 6, n))]
    else:
        li = rng.integers(0, len(lens), n)
        pick = [(lens[int(a)], int(k), int(j)) for a, k, j in zip(li, rng.integers(0, 6, n), rng.integers(0, 6, n))]
    fbs = [p[0] for p in pick]
    desc, sym_bytes, out_bytes = V.make_descs(fbs)
    sym = np.empty(sym_bytes, np.uint8)
    want = np.empty(out_bytes, np.uint8)
    for d, (fb, k, j) in zip(desc, pick):
        s, w = pool[(fb, k)]
        sym[int(d["sym_offset"]):int(d["sym_offset"]) + s.shape[1]] = s[j]
        want[int(d["out_offset"]):int(d["out_offset"]) + fb // 8] = w[j]
    d_sym = torch.from_numpy(sym).cuda()
    d_out = torch.full((out_bytes,), 0xEE, dtype=torch.uint8, device="cuda")
    if uniform:
        V.decode_batch_dev(d_sym, d_out, fbs[0], n)
    else:
        V.decode_varlen_dev(d_sym, d_out, torch.from_numpy(desc.view(np.uint8)).cuda(), n, max(fbs))
    torch.cuda.synchronize()
    nb = int((d_out.cpu().numpy() != want).sum())
    bad += nb
    total += n
    print(json.dumps({"table": seed, "entry": "batch" if uniform else "varlen", "frames": n, "lengths": lens, "differing_bytes": nb}), flush=True)
print(json.dumps({"tables": ntab, "total_frames": total, "differing_bytes": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
