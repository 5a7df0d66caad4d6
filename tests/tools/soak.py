#!/usr/bin/env python3
"""One-off soak: many frames, GPU (auto kernel) vs the CPU oracle, bit-exact.  Run on the GPU box."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize()
V.set_renorm_ge(0)  # the oracle's default comparator (`> 150`, the C decoders); the library's default is the MASM decoders' `>= 150`
dev = torch.device("cuda", 0)
ncpu = len(os.sched_getaffinity(0))
total = bad = 0
t0 = time.time()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # independent repetitions with different seeds
for rep, (fb, n) in [(r, c) for r in range(REPS) for c in ((768, 262144), (288, 131072), (1536, 65536), (3072, 32768), (6912, 16384), (9216, 8192), (776, 32768), (784, 32768), (770, 16384), (9214, 4096))]:
    for kind in ("noisy3dB", "noisy0dB", "noisy12dB", "uniform", "stress"):
        if kind == "uniform":
            sym = torch.randint(0, 256, (n, 4 * (fb + 6)), dtype=torch.uint8, device=dev, generator=torch.Generator(device=dev).manual_seed(fb + 100003 * rep))
        elif kind == "stress":
            # saturation / renormalisation-floor patterns: constant 0 / 255, random hard 0/255, 64-symbol runs
            g = torch.Generator(device=dev).manual_seed(fb + 7 + 100003 * rep)
            m = min(n, 4096)
            sl = 4 * (fb + 6)
            sym = torch.empty((m, sl), dtype=torch.uint8, device=dev)
            sym[0::4] = 0
            sym[1::4] = 255
            sym[2::4] = torch.randint(0, 2, (sym[2::4].shape[0], sl), dtype=torch.uint8, device=dev, generator=g) * 255
            runs = torch.randint(0, 256, (sym[3::4].shape[0], sl // 64 + 1), dtype=torch.uint8, device=dev, generator=g)
            sym[3::4] = runs.repeat_interleave(64, dim=1)[:, :sl]
        else:
            db = {"noisy3dB": 3.0, "noisy0dB": 0.0, "noisy12dB": 12.0}[kind]
            sym = make_frames(n, fb, seed=fb + len(kind) + 100003 * rep, device=dev, ebn0_db=db)
        n_k = sym.shape[0]
        out = torch.zeros((n_k, (fb + 7) // 8), dtype=torch.uint8, device=dev)
        V.decode_batch_dev(sym, out, fb, n_k); torch.cuda.synchronize()
        ref = O.decode_batch(fb, sym.cpu().numpy(), nthreads=ncpu, avx2=O.has_avx2() and (fb % 8 == 0))
        nb = int((out.cpu().numpy() != ref).any(axis=1).sum())
        total += n_k; bad += nb
        print(json.dumps({"rep": rep, "framebits": fb, "kind": kind, "frames": n_k, "differing": nb}), flush=True)
print(json.dumps({"total_frames": total, "differing": bad, "seconds": round(time.time() - t0, 1)}))

# ---- RS(120,110): random superframes with 0..8 symbol errors per column (clean, corrected, uncorrectable,
# miscorrected towards the virtual padding), GPU vs oracle: return values and every output byte ----
t0 = time.time()
rs_total = rs_bad = 0
for rep, (rsdims, nsf) in [(r, c) for r in range(REPS) for c in ((24, 6000), (12, 6000), (5, 4000), (1, 3000), (37, 2000), (256, 300), (300, 200), (8, 3000), (3, 3000))]:
    rng = np.random.default_rng(1000 + rsdims + 7919 * rep)
    msg = rng.integers(0, 256, (nsf * rsdims, 110), dtype=np.uint8)
    cws = np.stack([O.rs_encode(m) for m in msg[:512]])            # 512 distinct codewords, reused
    cw = cws[rng.integers(0, 512, nsf * rsdims)]
    # three regimes by superframe: correctable only / heavy incl. uncorrectable / mostly clean with rare 6..8
    regime = (np.arange(nsf * rsdims) // rsdims) % 3
    ne = np.where(regime == 0, rng.choice([0, 0, 0, 0, 1, 1, 2, 3, 4, 5], nsf * rsdims),
                  np.where(regime == 1, rng.choice([0, 0, 0, 1, 1, 2, 3, 4, 5, 6, 7, 8], nsf * rsdims),
                           rng.choice([0] * 60 + [1, 1, 2, 6, 7, 8], nsf * rsdims)))
    for i in np.nonzero(ne)[0]:
        pos = rng.choice(120, ne[i], replace=False)
        cw[i, pos] ^= rng.integers(1, 256, ne[i], dtype=np.uint8)
    p = cw.reshape(nsf, rsdims, 120).transpose(0, 2, 1).reshape(nsf, 120 * rsdims).copy()
    init = rng.integers(0, 256, (nsf, 110 * rsdims), dtype=np.uint8)
    ret_ref, out_ref = O.rs_check_batch(p, rsdims, out_init=init)
    d_out = torch.from_numpy(init.copy()).to(dev); d_ret = torch.zeros(nsf, dtype=torch.int32, device=dev)
    V.rs_batch_dev(torch.from_numpy(p).to(dev), d_out, d_ret, rsdims, nsf); torch.cuda.synchronize()
    nb = int((d_ret.cpu().numpy() != ret_ref).sum()) + int((d_out.cpu().numpy() != out_ref).any(axis=1).sum())
    rs_total += nsf; rs_bad += nb
    print(json.dumps({"rep": rep, "rsdims": rsdims, "superframes": nsf, "failed_superframes": int((ret_ref < 0).sum()), "differing": nb}), flush=True)
print(json.dumps({"rs_superframes": rs_total, "rs_differing": rs_bad, "seconds": round(time.time() - t0, 1)}))
