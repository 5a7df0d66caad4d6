#!/usr/bin/env python3
"""RS(120,110) batch rate on DISTINCT superframes (bench_rs.py tiles 64 distinct ones over the batch, so its input reads can be
served by the caches).  The code is linear: the XOR of three valid codeword blocks is a valid codeword block, which gives
64^3 distinct clean superframes from 64 oracle-encoded ones without encoding on the GPU.  Errors are injected on the GPU (one
symbol in a fraction of the columns); expected result: the clean rows back, return value = number of damaged columns.
usage: bench_rs_distinct.py [rsdims] [nsf] [fraction_of_columns_with_one_error]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import _vitpkg
V = _vitpkg.load_package(); O = _vitpkg.load_oracle(); V.initialize()
dev = torch.device("cuda", 0)
rsdims = int(sys.argv[1]) if len(sys.argv) > 1 else 24
nsf = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rng = np.random.default_rng(rsdims)
base = np.empty((64, 120, rsdims), np.uint8)
for s in range(64):
    for j in range(rsdims):
        base[s, :, j] = O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8))
d_base = torch.from_numpy(base.reshape(64, -1)).to(dev)
g = torch.Generator(device=dev); g.manual_seed(5)
i = torch.arange(nsf, device=dev)
d_clean = d_base[i % 64] ^ d_base[(i // 64) % 64] ^ d_base[(i // 4096) % 64]     # nsf distinct valid blocks (nsf <= 262144)
d_p = d_clean.clone()
ncol = nsf * rsdims
nerr = int(ncol * frac)
exp_ret = torch.zeros(nsf, dtype=torch.int32, device=dev)
if nerr:
    cols = torch.randperm(ncol, generator=g, device=dev)[:nerr]                     # distinct columns: one error each
    sf, col = cols // rsdims, cols % rsdims
    row = torch.randint(0, 120, (nerr,), generator=g, device=dev)
    val = torch.randint(1, 256, (nerr,), generator=g, device=dev, dtype=torch.int32).to(torch.uint8)
    flat = d_p.view(-1)
    pos = sf * (120 * rsdims) + row * rsdims + col
    flat[pos] ^= val
    exp_ret.index_add_(0, sf, torch.ones(nerr, dtype=torch.int32, device=dev))
d_out = torch.zeros((nsf, 110 * rsdims), dtype=torch.uint8, device=dev); d_ret = torch.zeros(nsf, dtype=torch.int32, device=dev)
te = time.perf_counter() + 0.06
while time.perf_counter() < te:
    V.rs_batch_dev(d_p, d_out, d_ret, rsdims, nsf); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): V.rs_batch_dev(d_p, d_out, d_ret, rsdims, nsf)
b.record(); torch.cuda.synchronize(); ms = a.elapsed_time(b) / 10
ok = bool(torch.equal(d_out, d_clean[:, :110 * rsdims])) and bool(torch.equal(d_ret, exp_ret))
# spot-check the expectation itself against the oracle on a few superframes
k = 16
r_ref, o_ref = O.rs_check_batch(d_p[:k].cpu().numpy(), rsdims)
ok_oracle = bool(np.array_equal(r_ref, d_ret[:k].cpu().numpy())) and bool(np.array_equal(o_ref, d_out[:k].cpu().numpy()))
print(json.dumps({"case": "RS on distinct superframes", "rsdims": rsdims, "nsf": nsf, "columns_with_one_error": frac, "ms": round(ms, 4),
                  "GB_s": round(nsf * 230 * rsdims / ms / 1e6, 1), "hbm_frac": round(nsf * 230 * rsdims / ms / 1e6 / 8000.0, 3),
                  "all_outputs_as_expected": ok, "oracle_spot_check_ok": ok_oracle}))
