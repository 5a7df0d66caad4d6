"""Per-group phase times of the long-frame kernel (diagnostic build -DVIT_DIAG_TIMES in VITERBI_AMD_LIB): forward pass (with the
in-flight parts inside), the in-flight parts alone (round-4 builds; the round-3 build reports the hardware slot there), the phase
after the forward pass.  usage: long_phases.py framebits frames [round3]"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize(); V.set_kernel(2)
lib = ctypes.CDLL(os.environ["VITERBI_AMD_LIB"])
fb, frames = int(sys.argv[1]), int(sys.argv[2])
r3 = len(sys.argv) > 3
dev = torch.device("cuda", 0)
sym = make_frames(frames, fb, seed=3, device=dev)
out = torch.zeros((frames, fb // 8), dtype=torch.uint8, device=dev)
for _ in range(10):
    V.decode_batch_dev(sym, out, fb, frames)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); V.decode_batch_dev(sym, out, fb, frames); b.record(); torch.cuda.synchronize()
buf = np.zeros(16384 * 4, np.uint64)
rc = lib.vit_diag_times(buf.ctypes.data_as(ctypes.c_void_p))
n = min(frames // 4, 16384)
t = buf.reshape(-1, 4)[:n].astype(np.int64)
tick = 0.01
fwd, end = (t[:, 1] - t[:, 0]) * tick, (t[:, 2] - t[:, 1]) * tick
res = {"lib": os.path.basename(os.environ["VITERBI_AMD_LIB"]), "framebits": fb, "frames": frames, "kernel_us": round(a.elapsed_time(b) * 1e3, 1),
       "group_us_mean": round(float((fwd + end).mean()), 1), "forward_us_mean": round(float(fwd.mean()), 1), "after_forward_us_mean": round(float(end.mean()), 1),
       "last_end_us": round(float((t[:, 2].max() - t[:, 0].min()) * tick), 1)}
if not r3:
    tr = t[:, 3] * tick
    parts = (fb + 255) // 256 - 1
    res["in_flight_us_mean"] = round(float(tr.mean()), 1)
    res["in_flight_us_per_part"] = round(float(tr.mean()) / max(parts, 1), 2)
print(json.dumps(res))
