"""Per-wave timeline of one launch of the headline batch (diagnostic build -DVIT_DIAG_TIMES): start, traceback start and end
of every workgroup (s_memtime, 100 MHz) -> rounds, spread, drain.  usage: python tools/exp/timeline.py [frames] [noisy|random] [framebits]"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ["VITERBI_AMD_LIB"] = os.path.join(ROOT, "tools", "exp", "libviterbi_times.so")
import _vitpkg
from bench import make_frames
V = _vitpkg.load_package(); V.initialize()
lib = ctypes.CDLL(os.environ["VITERBI_AMD_LIB"])
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
kind = sys.argv[2] if len(sys.argv) > 2 else "noisy"
dev = torch.device("cuda", 0)
fb = int(sys.argv[3]) if len(sys.argv) > 3 else 768
if kind == "noisy":
    sym = make_frames(frames, fb, seed=3, device=dev)
else:
    sym = torch.randint(0, 256, (frames, 4 * (fb + 6)), dtype=torch.uint8, device=dev)
out = torch.zeros((frames, fb // 8), dtype=torch.uint8, device=dev)
for _ in range(40):
    V.decode_batch_dev(sym, out, fb, frames)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); V.decode_batch_dev(sym, out, fb, frames); b.record(); torch.cuda.synchronize()
buf = np.zeros(16384 * 4, np.uint64)
rc = lib.vit_diag_times(buf.ctypes.data_as(ctypes.c_void_p))
n = min(frames // 4, 16384)
t = buf.reshape(-1, 4)[:n].astype(np.int64)
print(json.dumps({"rc": rc, "raw_min": [int(x) for x in t.min(axis=0)], "raw_max": [int(x) for x in t.max(axis=0)]}), flush=True)
if rc != 0 or t[:, 0].min() <= 0 or t[:, 2].max() - t[:, 0].min() > 10**9:
    sys.exit("timeline: implausible time stamps (rc %d)" % rc)
t0 = t[:, 0].min()
tick = 0.01  # us per s_memtime tick (100 MHz)
st, tb, en = (t[:, 0] - t0) * tick, (t[:, 1] - t0) * tick, (t[:, 2] - t0) * tick
order = np.argsort(st)
res = {"frames": frames, "input": kind, "rc": rc, "kernel_us_events": round(a.elapsed_time(b) * 1e3, 1), "waves": int(n),
       "last_end_us": round(float(en.max()), 1),
       "wave_duration_us": {"mean": round(float((en - st).mean()), 1), "p5": round(float(np.percentile(en - st, 5)), 1), "p95": round(float(np.percentile(en - st, 95)), 1)},
       "traceback_us": {"mean": round(float((en - tb).mean()), 2), "p95": round(float(np.percentile(en - tb, 95)), 2)},
       "start_us_percentiles": {str(p): round(float(np.percentile(st, p)), 1) for p in (0, 5, 24, 26, 49, 51, 74, 76, 95, 100)},
       "end_us_percentiles": {str(p): round(float(np.percentile(en, p)), 1) for p in (0, 5, 25, 50, 75, 90, 95, 99, 100)}}
# how many waves are resident over time (4096 slots): sample the occupancy every 5 us
grid = np.arange(0, min(float(en.max()) + 5, 5000.0), 5.0)
occ = [(int(((st <= g) & (en > g)).sum())) for g in grid]
res["resident_waves_every_5us"] = occ
print(json.dumps(res))
