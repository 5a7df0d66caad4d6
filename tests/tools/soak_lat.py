#!/usr/bin/env python3
"""Soak of the latency kernel (vit_set_kernel(3)) and of the automatic choice on small launches: random even lengths
2..9216, uniform and descriptor-table batches, u8 and u32 symbols, reference-style noise / uniform bytes / stress
patterns; every output byte compared with the oracle.  Run on the GPU box."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize()
V.set_renorm_ge(0)  # the oracle's default comparator (`> 150`, the C decoders); the library's default is the MASM decoders' `>= 150`
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0 = time.time(); frames = 0; bad = 0; cases = 0


def symbols(n, fb):
    k = rng.integers(0, 3)
    if k == 0:
        return O.noisy_frames(n, fb, seed=int(rng.integers(1, 1 << 30)))
    if k == 1:
        return O.uniform_symbols(n * O.sym_len(fb), seed=int(rng.integers(1, 1 << 30))).reshape(n, -1)
    s = rng.integers(0, 2, (n, O.sym_len(fb)), dtype=np.uint8) * 255
    s[0] = 0 if rng.integers(0, 2) else 255
    return s


while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else time.time() - t0 < 60:
    fb = int(2 * rng.integers(1, 4609)) if rng.random() < 0.3 else int(rng.choice([8, 96, 288, 768, 770, 778, 780, 1536, 3072]))
    n = int(rng.integers(1, 65 if fb <= 3072 else 9))
    sym = symbols(n, fb)
    want = O.decode_batch(fb, sym, nthreads=8)
    for kernel, u32 in ((3, False), (3, True), (0, False)):
        old = V.set_kernel(kernel)
        try:
            d_out = torch.full((n, (fb + 7) // 8), 0xEE, dtype=torch.uint8, device="cuda")
            if u32:
                V.decode_batch_dev_u32(torch.from_numpy(sym.astype(np.int32)).cuda(), d_out, fb, n)
            else:
                V.decode_batch_dev(torch.from_numpy(sym).cuda(), d_out, fb, n)
            torch.cuda.synchronize()
        finally:
            V.set_kernel(old)
        bad += int((d_out.cpu().numpy() != want).any(axis=1).sum())
        frames += n
    cases += 1
    # a descriptor table of mixed lengths through the same kernel
    fbs = [int(x) for x in 2 * rng.integers(1, 800, int(rng.integers(2, 40)))]
    desc, sb, ob = V.make_descs(fbs)
    s2 = O.uniform_symbols(sb, seed=int(rng.integers(1, 1 << 30)))
    w2 = np.concatenate([O.decode_batch(f, s2[int(d["sym_offset"]):int(d["sym_offset"]) + O.sym_len(f)])[0] for f, d in zip(fbs, desc)])
    old = V.set_kernel(3)
    try:
        d_out = torch.zeros(ob, dtype=torch.uint8, device="cuda")
        V.decode_varlen_dev(torch.from_numpy(s2).cuda(), d_out, torch.from_numpy(desc.view(np.uint8)).cuda(), len(fbs), max(fbs))
        torch.cuda.synchronize()
    finally:
        V.set_kernel(old)
    bad += int(not np.array_equal(d_out.cpu().numpy(), w2))
    frames += len(fbs)
print(json.dumps({"cases": cases, "frames": frames, "differing": bad, "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
