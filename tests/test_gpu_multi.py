"""Multi-GPU entry points on the one-GPU box: the C-ABI pipeline (vit_decode_stream_multi) with its RCCL loop-back
rank, sharding.decode_stream with the HIP decoder on backend "nccl" (world_size 1), and bench.py through its own
spawn path.  Runs with N > 1 belong to the driver's 8-GPU node; the plumbing for N > 1 is covered on gloo by
tests/test_shard_gloo.py and tests/test_bench_launcher.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames(O, n, fb, seed):
    a = O.noisy_frames(n - n // 3, fb, seed=seed)
    b = O.uniform_symbols((n // 3) * O.sym_len(fb), seed=seed + 1).reshape(n // 3, -1)
    return np.concatenate([a, b])


@pytest.mark.parametrize("fb,n,chunk,rootf,flags", [
    (768, 1000, 128, -1, 0),      # one device, no RCCL: chunks decoded in place
    (768, 1000, 128, -1, 1),      # + loop-back rank: every second block goes through ncclSend/ncclRecv and back
    (768, 997, 100, 30, 1),       # weighted root, ragged last chunk
    (768, 64, 100, 0, 1),         # the root only distributes
    (770, 333, 64, -1, 1),        # partial last byte: (framebits+7)/8 output bytes per frame
    (3072, 150, 16, -1, 1),       # long frames (spill kernel) on both ranks
])
def test_decode_stream_multi_matches_oracle(V, O, torch_cuda, fb, n, chunk, rootf, flags):
    torch = torch_cuda
    sym = _frames(O, n, fb, seed=fb + n)
    want = O.decode_batch(fb, sym, nthreads=8)
    d_sym = torch.from_numpy(sym).cuda()
    d_out = torch.full((n, (fb + 7) // 8), 0xEE, dtype=torch.uint8, device="cuda")
    for _ in range(2):  # second call: cached communicator, streams and buffers
        d_out.fill_(0xEE)
        torch.cuda.synchronize()
        V.decode_stream_multi(d_sym, d_out, fb, n, [torch.cuda.current_device()], chunk, rootf, flags)
        assert np.array_equal(d_out.cpu().numpy(), want)  # synchronous call: no device sync needed before the copy
    assert torch.cuda.current_device() == 0  # the caller's device is restored


def test_decode_stream_multi_error_between_group_start_and_end(V, O, torch_cuda):
    """fault injection on the loop-back path: an error return inside a send/recv group closes the group, ABORTS the
    communicators (no wait for transfers that may never complete), drops the context - and the next call rebuilds it
    and decodes correctly"""
    torch = torch_cuda
    fb, n = 768, 3000
    sym = _frames(O, n, fb, seed=5)
    want = O.decode_batch(fb, sym, nthreads=8)
    d_sym = torch.from_numpy(sym).cuda()
    d_out = torch.full((n, fb // 8), 0xEE, dtype=torch.uint8, device="cuda")
    dev = [torch.cuda.current_device()]
    V.decode_stream_multi(d_sym, d_out, fb, n, dev, 256, -1, V.MULTI_LOOPBACK)
    assert np.array_equal(d_out.cpu().numpy(), want)
    os.environ["VITERBI_AMD_TEST_MULTI_FAULT"] = "2"
    try:
        with pytest.raises(V.ViterbiError, match="injected fault"):
            V.decode_stream_multi(d_sym, d_out, fb, n, dev, 256, -1, V.MULTI_LOOPBACK)
    finally:
        del os.environ["VITERBI_AMD_TEST_MULTI_FAULT"]
    torch.cuda.synchronize()
    d_out.fill_(0xEE)
    V.decode_stream_multi(d_sym, d_out, fb, n, dev, 256, -1, V.MULTI_LOOPBACK)
    assert np.array_equal(d_out.cpu().numpy(), want)
    assert torch.cuda.current_device() == 0


def test_decode_stream_multi_rejects_bad_arguments(V, torch_cuda):
    torch = torch_cuda
    d = torch.zeros(4 * 774 * 4, dtype=torch.uint8, device="cuda")
    o = torch.zeros(96 * 4, dtype=torch.uint8, device="cuda")
    for devs, chunk, rootf, flags, fb in (([0, 0], 2, -1, 0, 768), ([99], 2, -1, 0, 768), ([0], 0, -1, 0, 768),
                                          ([0], 2, 0, 0, 768), ([0], 2, -1, 8, 768), ([0], 2, -1, 0, 769)):
        with pytest.raises(V.ViterbiError):
            V.decode_stream_multi(d, o, fb, 4, devs, chunk, rootf, flags)
    V.decode_stream_multi(d, o, 768, 0, [0], 2)  # nothing to do is not an error


def test_sharding_pipeline_with_hip_decoder_on_nccl(V, O, torch_cuda):
    """sharding.decode_stream on backend nccl (= RCCL), world_size 1, the HIP decoder as the per-rank decoder"""
    code = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import _vitpkg
from importlib import import_module
V = _vitpkg.load_package(); O = _vitpkg.load_oracle()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sh = import_module("viterbi_dll_amd.sharding")
fb, n = 768, 5000
sym = np.concatenate([O.noisy_frames(n - 100, fb, seed=1), O.uniform_symbols(100 * O.sym_len(fb), seed=2).reshape(100, -1)])
d_sym = torch.from_numpy(sym).cuda(); d_out = torch.zeros((n, fb // 8), dtype=torch.uint8, device="cuda")
def dec(s, o): V.decode_batch_dev(s, o, fb, s.shape[0])
plan = sh.decode_stream(d_sym, d_out, n, fb, dec, 512)
torch.cuda.synchronize()
ok = np.array_equal(d_out.cpu().numpy(), O.decode_batch(fb, sym, nthreads=8))
dist.barrier(); dist.destroy_process_group()
print("RESULT", ok, plan.nchunks)
''' % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "RESULT True 10" in p.stdout, p.stdout + p.stderr[-2000:]


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu",
                        "--prewarm-ms", "0"] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_through_its_spawn_path(torch_cuda):
    """`bench.py --gpus 1 --spawn`: the same child-process launch the bare `--gpus N` uses, backend nccl, HIP decoder"""
    r = _bench(["--gpus", "1", "--spawn", "--frames", "8192"])
    assert r["n_gpus"] == 1 and r["config"]["launch"] == "spawned by bench.py" and r["value"] > 1000
    # the second stage of the path (RScheckSuperframe in batch) rides along at N = 1, checked against its construction
    ss = r["second_stage"]
    assert ss["outputs_and_return_values_as_constructed"] is True and ss["roofline"]["achieved"] > 100, ss
    # a plain N = 1 run also carries the two-stream leg next to `value`, with both of its output buffers equal
    r1 = _bench(["--frames", "8192", "--no-rs"])
    assert r1["pipelined"]["streams"] == 2 and r1["pipelined"]["both_output_buffers_equal"] is True and r1["pipelined"]["value"] > 1000
    assert "pipelined" not in _bench(["--frames", "8192", "--no-rs", "--no-pipelined"])
    r = _bench(["--gpus", "1", "--spawn", "--frames", "8192", "--mode", "scatter", "--chunk-frames", "1024"])
    assert r["n_gpus"] == 1 and r["value"] > 1000
    r = _bench(["--gpus", "1", "--frames", "8192", "--mode", "multi", "--chunk-frames", "1024", "--loopback"])
    assert r["n_gpus"] == 1 and r["value"] > 500 and r["multi_matches_single_launch"] is True


def test_device_index_out_of_range_fails_loudly(torch_cuda):
    """VITERBI_AMD_DEVICE beyond the usable gfx950 devices: no silent fall-back to another GPU (round-1 advisor finding)"""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, _vitpkg; V = _vitpkg.load_package();"
            "rc, out = V.deconvolve(768, np.full(4 * 774, 128, np.uint32));"
            "print('RESULT', rc, V.device_count(), V.GetCPUCaps(), V.last_error())" % ROOT)
    env = dict(os.environ, VITERBI_AMD_DEVICE="63")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0]
    assert line.startswith("RESULT 1 ") and "out of range" in line, line
