#!/usr/bin/env python3
"""Soak of the variable-length entry point (TEST INFRASTRUCTURE: the CPU oracle is the checker).

Random descriptor tables - DAB sizes, the lengths around the single-segment / long-frame boundary (776 ... 784), partial last
bytes, a few long frames among many short ones (split between the kernels) and the other way round, table sizes from one
round of workgroups to several - decoded by vit_decode_varlen_dev (auto kernel) and compared byte for byte with the oracle.
usage: python tests/tools/soak_varlen.py [tables]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize()
V.set_renorm_ge(0)  # the oracle's default comparator (`> 150`, the C decoders); the library's default is the MASM decoders' `>= 150`
ntab = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
total = bad = 0
t0 = time.time()
for seed in range(ntab):
    rng = np.random.default_rng(9000 + seed)
    kind = seed % 4
    n = int(rng.choice([300, 5000, 16384, 16400, 40000, 70000]))
    dab = 96 * rng.integers(3, 73, n)
    edge = rng.choice([768, 770, 776, 778, 780, 782, 784, 786, 1566, 1568], n)
    if kind == 0:    # config-3 like mix
        fbs = dab
    elif kind == 1:  # mostly FIC-sized, a few long ones: the table is split between the kernels
        fbs = np.where(rng.random(n) < 0.03, 96 * rng.integers(9, 73, n), edge)
    elif kind == 2:  # mostly long
        fbs = np.where(rng.random(n) < 0.9, 96 * rng.integers(9, 73, n), edge)
    else:            # any even length
        fbs = 2 * rng.integers(1, 2400, n)
    fbs = fbs.astype(np.int64).tolist()
    desc, sym_bytes, out_bytes = V.make_descs(fbs)
    # symbols: reference-style noise for a pool of frames per length would be slow; uniform bytes + hard decisions exercise the
    # re-trace passes and the saturation paths, a third of the table gets clean (noise-free) encoded frames via the oracle
    sym = O.uniform_symbols(sym_bytes, seed=seed)
    hard = rng.random(len(fbs)) < 0.3
    for i in np.nonzero(hard)[0][:2000]:
        o, L = int(desc[i]["sym_offset"]), O.sym_len(fbs[i])
        sym[o:o + L] = np.where(sym[o:o + L] & 1, 255, 0)
    d_sym = torch.from_numpy(sym).to(dev)
    d_out = torch.full((out_bytes,), 0xEE, dtype=torch.uint8, device=dev)
    d_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    V.decode_varlen_dev(d_sym, d_out, d_desc, len(fbs), max(fbs))
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    # oracle: batch per distinct length
    want = np.full(out_bytes, 0xEE, np.uint8)
    order = np.argsort(np.asarray(fbs), kind="stable")
    k = 0
    while k < len(order):
        fb = fbs[order[k]]
        j = k
        while j < len(order) and fbs[order[j]] == fb:
            j += 1
        idx = order[k:j]
        L, nb = O.sym_len(fb), (fb + 7) // 8
        frames = np.stack([sym[int(desc[i]["sym_offset"]):int(desc[i]["sym_offset"]) + L] for i in idx])
        dec = O.decode_batch(fb, frames, nthreads=8)
        for r, i in enumerate(idx):
            oo = int(desc[i]["out_offset"])
            want[oo:oo + nb] = dec[r]
        k = j
    nbad = int((got != want).sum())
    total += len(fbs); bad += nbad
    print(json.dumps({"table": seed, "kind": kind, "frames": len(fbs), "max_framebits": max(fbs), "differing_bytes": nbad}), flush=True)
print(json.dumps({"tables": ntab, "total_frames": total, "differing_bytes": bad, "seconds": round(time.time() - t0, 1)}))
