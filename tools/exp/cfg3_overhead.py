"""Config 3 (32768 frames, 288 ... 6912 bits): the descriptor entry as called (device sort, split gate, side stream) against the same table
sorted on the HOST and launched with VITERBI_AMD_NO_SORT=1 (no sort launches, no second kernel, no fork/join): the launcher's overhead."""
import json, os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) == 1:
    for mode in ("device_sort", "host_sorted"):
        env = dict(os.environ)
        if mode == "host_sorted":
            env["VITERBI_AMD_NO_SORT"] = "1"
        subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch, _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize(); V.set_kernel(2)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
n = 32768
fbs = 96 * rng.integers(3, 73, n)
if sys.argv[1] == "host_sorted":
    fbs = np.sort(fbs)[::-1].copy()
desc, sym_bytes, out_bytes = V.make_descs(fbs.tolist())
sym = torch.empty(sym_bytes, dtype=torch.uint8, device=dev)
so_all = torch.from_numpy(desc["sym_offset"].astype(np.int64)).to(dev)
for m in range(3, 73):
    idx = torch.from_numpy(np.nonzero(fbs == 96 * m)[0]).to(dev)
    if idx.numel():
        fr = make_frames(int(idx.numel()), 96 * m, seed=300 + m, device=dev)
        pos = so_all[idx][:, None] + torch.arange(fr.shape[1], device=dev)[None, :]
        sym[pos.reshape(-1)] = fr.reshape(-1)
d_desc = torch.from_numpy(desc.view(np.uint8)).to(dev)
out = torch.zeros(out_bytes, dtype=torch.uint8, device=dev)
fn = lambda: V.decode_varlen_dev(sym, out, d_desc, n, int(fbs.max()))
for _ in range(30): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
print(json.dumps({"mode": sys.argv[1], "ms": round(a.elapsed_time(b) / 20, 4), "Gbit_s": round(float(fbs.sum()) / (a.elapsed_time(b) / 20) / 1e6, 1)}), flush=True)
