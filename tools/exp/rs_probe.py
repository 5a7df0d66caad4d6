import json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames, rs_encode_columns  # the encoder is bench.py's own (no oracle outside tests/)
V = _vitpkg.load_package(); V.initialize()
dev = torch.device("cuda", 0)
rsdims, nsf, base_n = 24, 16384, 64
fb = 192 * rsdims
rng = np.random.default_rng(500 + rsdims)
cw = rs_encode_columns(rng.integers(0, 256, (110, base_n * rsdims), dtype=np.uint8))  # (120, base_n*rsdims)
blocks = cw.reshape(120, base_n, rsdims).transpose(1, 0, 2).copy()
bits = np.unpackbits(blocks.reshape(base_n, -1), axis=1).reshape(base_n * 5, fb)
pb = torch.from_numpy(bits.astype(np.int32)).to(dev).repeat(nsf // base_n, 1)
sym = make_frames(nsf * 5, fb, seed=7, device=dev, payload_bits=pb)
d_work = torch.zeros((nsf, 120 * rsdims), dtype=torch.uint8, device=dev)
d_out = torch.zeros((nsf, 110 * rsdims), dtype=torch.uint8, device=dev)
d_ret = torch.zeros(nsf, dtype=torch.int32, device=dev)
V.decode_batch_dev(sym, d_work, fb, nsf * 5); torch.cuda.synchronize()
clean = torch.from_numpy(blocks.reshape(base_n, -1)).to(dev).repeat(nsf // base_n, 1)
diff = (d_work != clean).view(nsf, 120, rsdims)
per_col = diff.sum(dim=1)  # errors per column
hist = torch.bincount(per_col.view(-1), minlength=8).cpu().numpy()
print("errors per column histogram:", hist.tolist())
def timeit(fn, n=20):
    te = time.perf_counter() + 0.06
    while time.perf_counter() < te: fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n
print("rs on decoded data ms", round(timeit(lambda: V.rs_batch_dev(d_work, d_out, d_ret, rsdims, nsf)), 4))
print("rs on clean data   ms", round(timeit(lambda: V.rs_batch_dev(clean, d_out, d_ret, rsdims, nsf)), 4))
# waves (64 consecutive columns) that contain a column with >= 2 errors
w = (per_col.view(-1, 64) >= 2).any(dim=1).float().mean().item()
print("fraction of 64-column waves with a multi-error column:", round(w, 3))
