#!/usr/bin/env python3
"""bench.py -- headline benchmark: decoded Mbit/s on batched 768-bit DAB FIC frames.

One "step" = one pass of the hot path (K=7 r=1/4 Viterbi: ACS + traceback) over one
batch of 65536 FIC frames (BASELINE.json configs[1]) that is already resident in HBM
in the device format (1 byte per soft symbol).  N>1 = one process per GPU, every rank
decodes its own 65536-frame shard (independent frames, no data-path collective: weak
scaling); value = all ranks' decoded bits / max-over-ranks time.

Launching: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself (fresh child processes, created before this process has touched the GPU); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line (rank 0).  `roofline` prices the decode kernel against the HBM
peak with the ALGORITHMIC bytes (3192 B per FIC frame: 3096 symbol bytes in + 96 out);
`cpu_baseline` is this repo's own AVX2 port of the same integer specification
(oracle/vit_avx2.c, "port") timed on the host cores of the same box, on the same frames.
"""
import argparse
import json
import os
import random
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _vitpkg  # noqa: E402

FRAMEBITS = 768
TAIL = 6
POLYS = (109, 79, 83, 109)  # viterbi-benchmark.cpp:64
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_frames(nframes, framebits, seed, device, ebn0_db=3.0, return_bits=False, payload_bits=None):
    """Reference-style synthetic input (viterbi-benchmark.cpp:293-311): random bits ->
    DAB mother code -> AWGN at Eb/N0 = 3 dB, sample = 127.5 + 32*N(+-gain,1), clip 0..255.
    Built on the GPU with a seeded torch generator; returns uint8 [nframes, 4*(framebits+6)].
    payload_bits (int32 [nframes, framebits], 0/1) replaces the random payload."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T = framebits + TAIL
    if payload_bits is not None:
        bits = payload_bits.to(device=device, dtype=torch.int32)
    else:
        bits = torch.randint(0, 2, (nframes, framebits), generator=g, device=device, dtype=torch.int32)
    bits = torch.cat([bits, torch.zeros((nframes, TAIL), dtype=torch.int32, device=device)], dim=1)
    # sr(t) = last 7 input bits, newest in bit 0
    padded = torch.cat([torch.zeros((nframes, 6), dtype=torch.int32, device=device), bits], dim=1)
    sr = torch.zeros((nframes, T), dtype=torch.int32, device=device)
    for k in range(7):
        sr |= padded[:, 6 - k:6 - k + T] << k
    hard = torch.empty((nframes, T, 4), dtype=torch.float32, device=device)
    for j, poly in enumerate(POLYS):
        x = sr & poly
        x = x ^ (x >> 4)
        x = x ^ (x >> 2)
        x = x ^ (x >> 1)
        hard[:, :, j] = (x & 1).float()
    esn0 = ebn0_db + 10.0 * np.log10(1.0 / 4.0)
    gain = 1.0 / np.sqrt(0.5 / 10.0 ** (esn0 / 10.0))
    noise = torch.randn((nframes, T, 4), generator=g, device=device, dtype=torch.float32)
    v = 127.5 + 32.0 * ((hard * 2.0 - 1.0) * gain + noise)
    sym = v.to(torch.int32).clamp_(0, 255).to(torch.uint8)  # C truncation then clip
    sym = sym.reshape(nframes, 4 * T).contiguous()
    return (sym, bits[:, :framebits]) if return_bits else sym


def cpu_baseline(O, sym_host, framebits, want_seconds=10.0):
    """AVX2 port on the host: 1 pinned thread (the x10 target's denominator) and all cores."""
    n = sym_host.shape[0]
    ncpu = len(os.sched_getaffinity(0))
    avx2 = O.has_avx2()
    # reference output for the parity check; the first pass also creates the threads and touches the pages,
    # so the all-cores figure is taken from a second, warm pass
    ref = O.decode_batch(framebits, sym_host, nthreads=ncpu, avx2=avx2)
    t0 = time.perf_counter()
    O.decode_batch(framebits, sym_host, nthreads=ncpu, avx2=avx2)
    t_all = time.perf_counter() - t0
    # single thread on a bounded sample sized for ~want_seconds
    t0 = time.perf_counter()
    O.decode_batch(framebits, sym_host[:2048], nthreads=1, avx2=avx2)
    per_frame = (time.perf_counter() - t0) / 2048
    ns = int(min(n, max(2048, want_seconds / per_frame)))
    reps = max(1, int(round(want_seconds / (per_frame * ns))))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.decode_batch(framebits, sym_host[:ns], nthreads=1, avx2=avx2)
    t1 = time.perf_counter() - t0
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    base = {
        "value": round(ns * reps * framebits / t1 / 1e6, 2),
        "unit": "Mbit/s",
        "cores": 1,
        "kind": "port",
        "impl": "oracle/vit_avx2.c (own AVX2 port)" if avx2 else "oracle/vit_oracle.c (scalar)",
        "sample": "%d FIC frames x %d passes, 1 thread (%.1f s)" % (ns, reps, t1),
        "all_cores": {"value": round(n * framebits / t_all / 1e6, 2), "cores": ncpu,
                      "sample": "%d frames, second (warm) pass" % n},
        "cpu": model,
    }
    return base, ref


# ---- second stage of the path: RScheckSuperframe in batch (BASELINE config 5's RS(120,110) step) ----------------------
def rs_encode_columns(msg):
    """msg: (110, ncol) bytes -> (120, ncol) RS(120,110) codewords, one per column, from the code's definition:
    GF(2^8)/0x11D, g(x) = prod_{i<10}(x + alpha^i), systematic (message first, then the remainder, x^9 first)."""
    alpha = np.zeros(512, np.int64); logt = np.zeros(256, np.int64)
    x = 1
    for i in range(255):
        alpha[i] = x; logt[x] = i
        x <<= 1
        if x & 256:
            x ^= 0x11D
    alpha[255:510] = alpha[:255]
    g = [1]
    for i in range(10):  # multiply by (x + alpha^i); g[j] = coefficient of x^j
        ng = [0] * (len(g) + 1)
        for j, c in enumerate(g):
            ng[j + 1] ^= c
            if c:
                ng[j] ^= int(alpha[logt[c] + i])
        g = ng
    glog = [int(logt[c]) for c in g[:10]]  # g is monic and none of g_0..g_9 is zero for this code
    msg = msg.astype(np.int64)
    r = np.zeros((10, msg.shape[1]), np.int64)  # remainder register, r[j] = coefficient of x^j
    for k in range(110):
        f = msg[k] ^ r[9]
        nz = f != 0
        lf = logt[f]
        nr = np.empty_like(r)
        for j in range(10):
            prod = np.where(nz, alpha[lf + glog[j]], 0)
            nr[j] = (r[j - 1] if j else 0) ^ prod
        r = nr
    return np.concatenate([msg, r[::-1]], axis=0).astype(np.uint8)


def rs_test_block(nsf, rsdims, p_err, seed):
    """nsf DISTINCT superframes - codeword byte k of column j at p[j + k*rsdims] (rschecksf.cpp:75-76) - with ONE symbol
    error in a fraction p_err of the columns (what the Viterbi stage leaves at Eb/N0 = 3 dB).  Returns the block, the
    expected output (the messages) and the expected return values (corrections per superframe): the check needs no oracle."""
    rng = np.random.default_rng(seed)
    ncol = nsf * rsdims
    cw = rs_encode_columns(rng.integers(0, 256, (110, ncol), dtype=np.int64))
    err = rng.random(ncol) < p_err
    rows = rng.integers(0, 120, ncol)
    vals = rng.integers(1, 256, ncol).astype(np.uint8)
    bad = cw.copy()
    cols = np.nonzero(err)[0]
    bad[rows[cols], cols] ^= vals[cols]
    to_block = lambda a, h: a.reshape(h, nsf, rsdims).transpose(1, 0, 2).reshape(nsf, h * rsdims).copy()
    return to_block(bad, 120), to_block(cw[:110], 110), err.reshape(nsf, rsdims).sum(axis=1).astype(np.int32)


def second_stage(V, dev, nsf=131072, rsdims=24, distinct=256, iters=100):
    p, want_out, want_ret = rs_test_block(distinct, rsdims, 0.06, seed=4242)
    reps = nsf // distinct
    d_p = torch.from_numpy(p).to(dev).repeat(reps, 1).contiguous()
    d_out = torch.zeros((nsf, 110 * rsdims), dtype=torch.uint8, device=dev)
    d_ret = torch.full((nsf,), -7, dtype=torch.int32, device=dev)
    t_end = time.perf_counter() + 0.15  # untimed pre-conditioning: the clocks have dropped during the CPU leg
    while time.perf_counter() < t_end:
        V.rs_batch_dev(d_p, d_out, d_ret, rsdims, nsf)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        V.rs_batch_dev(d_p, d_out, d_ret, rsdims, nsf)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ok = bool(torch.equal(d_out.view(reps, distinct, -1), torch.from_numpy(want_out).to(dev).expand(reps, -1, -1))) and \
        bool(torch.equal(d_ret.view(reps, distinct), torch.from_numpy(want_ret).to(dev).expand(reps, -1)))
    gbs = nsf * 230.0 * rsdims / (ms * 1e-3) / 1e9  # 120*RSDims read + 110*RSDims written per superframe
    return {"kernel": "rs_kernel: RScheckSuperframe in batch (vit_rs_batch_dev), RSDims %d" % rsdims,
            "workload": "%d superframes resident in HBM (%d distinct, tiled), one symbol error in 6 %% of the columns" % (nsf, distinct),
            "ms": round(ms, 4), "superframes_per_s": round(nsf / (ms * 1e-3), 0),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(nsf * 230 * rsdims)},
            "outputs_and_return_values_as_constructed": ok}


def precondition(launch, ms=150.0):
    """untimed: ~15 ms of sustained load bring the MI355X to its steady-state clocks; a leg that follows seconds of
    GPU idle (the CPU baseline) starts ~10 % slow without it (tools/exp/trend.py)"""
    t_end = time.perf_counter() + ms / 1e3
    k = 0
    while time.perf_counter() < t_end:
        for _ in range(8):
            launch(k)
            k += 1
        torch.cuda.synchronize()


def pipelined_leg(V, d_sym, d_out, launches=400):
    """`launches` steps (a fixed count, whatever --steps is) with consecutive launches ALTERNATING BETWEEN TWO HIP STREAMS
    (two output buffers): while one launch drains, the next one already fills the chip, so the fixed ~20 us a launch of
    this kernel spends ramping up and draining overlap with useful work.  What a host that streams batch after batch
    gets; reported NEXT TO `value`, which stays the one-stream figure (whose kernel duration is what the roofline
    object and the rocprof summary price).  Timed with HIP events on BOTH streams: start = one event both streams wait
    for, end = the later of the two streams' last launches."""
    n = d_sym.shape[0]
    outs = [d_out, torch.zeros_like(d_out)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def launch(k):
        V.decode_batch_dev(d_sym, outs[k & 1], FRAMEBITS, n, stream=streams[k & 1].cuda_stream)

    torch.cuda.synchronize()
    precondition(launch)
    e0 = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    e0.record(streams[0])
    streams[1].wait_event(e0)
    t0 = time.perf_counter()
    for k in range(launches):
        launch(k)
    ends[0].record(streams[0])
    ends[1].record(streams[1])
    torch.cuda.synchronize()
    host_ms = (time.perf_counter() - t0) * 1e3
    ms = max(e0.elapsed_time(ends[0]), e0.elapsed_time(ends[1]))
    same = bool(torch.equal(outs[0], outs[1]))
    return {"streams": 2, "steps": launches, "ms_per_step": round(ms / launches, 4),
            "value": round(n * FRAMEBITS * launches / (ms * 1e-3) / 1e6, 1) if same else 0.0, "unit": "Mbit/s",
            "timing": "HIP events on both streams (host clock over the same launches: %.4f ms per step)" % (host_ms / launches),
            "preconditioned_ms": 150.0, "both_output_buffers_equal": same,
            "what": "consecutive launches alternate between two HIP streams: ramp and drain of neighbouring launches overlap"}


def input_sensitivity(V, O, dev, n, seed, ge, launches=300):
    """The traceback is the one input-dependent part of the kernel: the same batch shape at Eb/N0 = 0 dB and on uniform
    random bytes (no signal at all: the worst case), `launches` launches each behind their own pre-conditioning, HIP
    events on the launch stream; every frame compared with the oracle AFTER the timed region (O = None: not checked)."""
    out = {}
    for name in ("0dB", "random_bytes"):
        if name == "0dB":
            d_sym = make_frames(n, FRAMEBITS, seed=seed + 17, device=dev, ebn0_db=0.0)
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(seed + 29)
            d_sym = torch.randint(0, 256, (n, 4 * (FRAMEBITS + TAIL)), generator=g, dtype=torch.uint8, device=dev)
        d_out = torch.zeros((n, (FRAMEBITS + 7) // 8), dtype=torch.uint8, device=dev)
        precondition(lambda k: V.decode_batch_dev(d_sym, d_out, FRAMEBITS, n))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            V.decode_batch_dev(d_sym, d_out, FRAMEBITS, n)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / launches
        rec = {"ms_per_step": round(ms, 4), "value": round(n * FRAMEBITS / (ms * 1e-3) / 1e6, 1), "launches": launches}
        if O is not None:
            ref = O.decode_batch(FRAMEBITS, d_sym.cpu().numpy(), nthreads=len(os.sched_getaffinity(0)), ge=bool(ge))
            bad = int((d_out.cpu().numpy() != ref).any(axis=1).sum())
            rec["frames_checked"], rec["frames_differing"] = n, bad
            if bad:
                rec["value"] = 0.0
        out[name] = rec
        del d_sym, d_out
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps (default 2000 = 0.9 s of back-to-back launches: long enough for an external GPU-busy "
                         "sampler to see the timed region; 20 steps measure the same rate within 1 %%, profiles/README.md)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=65536, help="FIC frames per GPU per step")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 wave-per-frame, 2 packed")
    ap.add_argument("--renorm-ge", type=int, default=0,
                    help="1: the MASM decoders' `>= 150` renormalise comparator (vit_set_renorm_ge) instead of the C "
                         "decoders' `> 150`; the parity check then uses the oracle's ge mode (scalar, not the AVX2 port)")
    ap.add_argument("--input", choices=["noisy", "random"], default="noisy",
                    help="noisy: reference-style symbols, Eb/N0 = 3 dB (the headline workload); random: uniform random bytes - "
                         "no signal at all, the traceback's worst case (tests/tools/bench_inputs.py)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-rs", action="store_true", help="skip the second-stage (RScheckSuperframe batch) measurement")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-stream leg (`pipelined` in the JSON line)")
    ap.add_argument("--no-sensitivity", action="store_true",
                    help="skip the input-family leg (`input_sensitivity`: the same batch at 0 dB and on uniform random bytes)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed GPU pre-conditioning before the W warm-up steps: the MI355X needs ~15 ms of sustained load "
                         "to reach its steady-state clocks (first launches run ~10 %% slower, tools/exp/trend.py)")
    ap.add_argument("--mode", choices=["shard", "scatter", "scatter-plain", "multi"], default="shard",
                    help="shard: every rank owns its frames (default, no collective); scatter: rank 0 owns all frames, "
                         "chunked + overlapped RCCL send/recv pipeline inside the timed step (config 4, "
                         "sharding.decode_stream); scatter-plain: ONE scatter + decode + ONE gather (f mod N); multi: ONE "
                         "process drives all --gpus devices through the C ABI (vit_decode_stream_multi, same pipeline)")
    ap.add_argument("--loopback", action="store_true",
                    help="multi mode: add the RCCL loop-back rank on the root device (self-test on a one-GPU box)")
    ap.add_argument("--chunk-frames", type=int, default=32768, help="scatter mode: frames per rank per chunk")
    ap.add_argument("--root-frames", type=int, default=None,
                    help="scatter mode: frames the root keeps per chunk (default = --chunk-frames)")
    ap.add_argument("--no-scatter-leg", action="store_true",
                    help="N > 1, shard mode: skip the extra leg that pushes the same frames through the one-root RCCL "
                         "scatter pipeline and reports it next to the shard value (`scatter` in the JSON line)")
    ap.add_argument("--scatter-timeout", type=int, default=150, help="seconds the scatter leg may take before it is abandoned")
    ap.add_argument("--link-gbs", type=float, default=60.0,
                    help="assumed practical xGMI rate per link and direction, for the scatter leg's split and its bound")
    ap.add_argument("--spawn", action="store_true", help="start the ranks as child processes even for --gpus 1")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo + --stub: CPU rehearsal of the launcher and the collectives (tests)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY: no GPU, the decoder is replaced by a byte-copy stand-in; the JSON line says so "
                         "and its value is not a measurement")
    return ap.parse_args(argv)


def _free_port():
    """A rendezvous port for ranks that start seconds from now.  NOT bind(0): that hands out a port of the kernel's ephemeral range
    (ip_local_port_range, 32768-60999 here), which any outgoing connection of any process may take before rank 0 listens on it (seen
    once: EADDRINUSE in the spawn test).  A random port BELOW that range that can be bound right now is only ever taken by another
    listener."""
    lo_eph = 32768
    try:
        lo_eph = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError, IndexError):
        pass
    hi = max(min(lo_eph, 32768) - 1, 12000)
    rng = random.Random(os.getpid() ^ int(time.time() * 1e6))
    for _ in range(64):
        port = rng.randint(10000, hi)
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            continue
        finally:
            s.close()
        return port
    s = socket.socket()  # last resort
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """The bare `python bench.py --gpus N`: start N ranks (one per GPU) as fresh child processes.  Nothing in
    this process has initialised HIP (importing torch does not), and it never will: it only waits.  The
    children inherit stdout/stderr, so rank 0's JSON line is this command's JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    env["WORLD_SIZE"] = str(args.gpus)
    env["VIT_BENCH_SPAWNED"] = "1"
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def stub_decode(sym_block, out_block):
    """TEST ONLY stand-in for the decoder (CPU rehearsal): the first bytes of every frame's symbols"""
    out_block.copy_(sym_block[:, :out_block.shape[1]])


def scatter_split(frames_per_s, world, n_total, link_gbs, framebits=FRAMEBITS):
    """one root feeds W-1 peers through one xGMI link each.  A peer can be fed min(D, L) frames/s
    (D = one GPU's decode rate, L = link rate in frames), the root decodes at D meanwhile: balanced when the root
    keeps D / min(D, L) times a peer's block.  Returns (chunk_frames, root_frames, bound) with about eight chunks in
    the stream so that the pipeline has something to overlap; bound = predicted speed-up over ONE GPU."""
    link_fps = link_gbs * 1e9 / (4.0 * (framebits + TAIL))
    peer_fps = min(frames_per_s, link_fps)
    ratio = frames_per_s / peer_fps
    chunk = int(n_total / (8.0 * (ratio + world - 1))) & ~3
    chunk = max(4, chunk)
    root = max(4, int(round(ratio * chunk)) & ~3)
    return chunk, root, {"decode_frames_per_s_per_gpu": round(frames_per_s, 0), "link_frames_per_s": round(link_fps, 0),
                         "link_GBs_per_direction_assumed": link_gbs,
                         "speedup_vs_1gpu_bound": round(1.0 + (world - 1) * peer_fps / frames_per_s, 2),
                         "bound": ("root xGMI egress: every peer is fed through ONE link (DESIGN.md (e))" if link_fps < frames_per_s
                                   else "the decoders: a link carries frames faster than one GPU decodes them")}


def scatter_leg(args, dist, sharding, rank, world, dev, d_sym, d_out, decode_into, sync, frames_per_s):
    """N > 1, run by EVERY rank right after the shard measurement: the same frames, now all owned by rank 0 and pushed
    through sharding.decode_stream (chunked RCCL send/recv pipeline) with the weighted split of DESIGN.md (e)."""
    n = d_sym.shape[0]
    n_total = n * world
    if args.stub:
        frames_per_s = 2.0 * args.link_gbs * 1e9 / (4.0 * (FRAMEBITS + TAIL))  # rehearsal: pretend D = 2 L
    chunk, rootf, bound = scatter_split(frames_per_s, world, n_total, args.link_gbs)
    d_all = d_sym.repeat(world, 1) if rank == 0 else None   # rank 0's shard, tiled: the expected output is known
    d_all_out = torch.zeros((n_total, d_out.shape[1]), dtype=torch.uint8, device=dev) if rank == 0 else None
    steps = max(1, min(args.steps, 5))

    def step():
        sharding.decode_stream(d_all, d_all_out, n_total, FRAMEBITS, decode_into, chunk, rootf)

    if os.environ.get("VIT_BENCH_TEST_HANG_SCATTER") and rank == world - 1:
        time.sleep(3600)  # TEST HOOK (tests/test_bench_launcher.py): one rank never reaches the collective - what a hung RCCL leg looks like
    step()  # warm-up: buffers, communicator channels
    sync()
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        return None
    want = d_sym[:, :d_out.shape[1]] if args.stub else d_out  # stub decoder = byte copy; GPU: the shard leg's output
    same = bool(torch.equal(d_all_out.view(world, n, -1), want.unsqueeze(0).expand(world, -1, -1)))
    return {"mode": "scatter: rank 0 owns all %d frames; sharding.decode_stream, chunked + overlapped RCCL send/recv "
                    "pipeline inside the step (BASELINE config 4 taken literally)" % n_total,
            "value": round(n_total * FRAMEBITS * steps / dt / 1e6, 1) if same else 0.0, "unit": "Mbit/s",
            "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "frames": n_total,
            "chunk_frames": chunk, "root_frames": rootf, "every_frame_matches_the_shard_decode": same,
            "predicted": bound}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.mode == "multi":
        return main_multi(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        rc = spawn_ranks(args, argv)
        if rc:
            sys.exit(rc)
        return None

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    spawned = "WORLD_SIZE" in os.environ
    dist = None
    if spawned:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner to stdout when its first communicator comes up: keep stdout for the JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.stub:
                dist.init_process_group(args.backend)
            else:
                os.environ.setdefault("VITERBI_AMD_DEVICE", str(local_rank))
                torch.cuda.set_device(local_rank)
                dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    elif not args.stub:
        torch.cuda.set_device(0)
    dev = torch.device("cpu") if args.stub else torch.device("cuda", local_rank if spawned else 0)
    sync = (lambda: None) if args.stub else torch.cuda.synchronize

    V = _vitpkg.load_package()
    if not args.stub:
        assert V.device_count() >= 1, "libviterbi.so sees no gfx950 device: " + V.last_error()
        V.initialize()
        V.set_kernel(args.kernel)
        V.set_renorm_ge(args.renorm_ge)

    n = args.frames
    out_len = (FRAMEBITS + 7) // 8
    d_sym = make_frames(n, FRAMEBITS, seed=1234 + rank, device=dev)
    if args.input == "random":
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        d_sym = torch.randint(0, 256, tuple(d_sym.shape), generator=g, dtype=torch.uint8, device=dev)
    d_out = torch.zeros((n, out_len), dtype=torch.uint8, device=dev)
    sync()

    def decode_into(sym_block, out_block):
        if args.stub:
            stub_decode(sym_block, out_block)
        else:
            V.decode_batch_dev(sym_block, out_block, FRAMEBITS, sym_block.shape[0])  # enqueues on torch's current stream

    routing_ok = None
    if args.mode != "shard" and dist:
        from importlib import import_module
        sharding = import_module("viterbi_dll_amd.sharding")
        n_total = n * world
        d_all = make_frames(n_total, FRAMEBITS, seed=99, device=dev) if rank == 0 else None
        d_all_out = torch.zeros((n_total, out_len), dtype=torch.uint8, device=dev) if rank == 0 else None
        if args.mode == "scatter":
            def step():
                sharding.decode_stream(d_all, d_all_out, n_total, FRAMEBITS, decode_into, args.chunk_frames,
                                       args.root_frames)
        else:
            d_loc_out = torch.zeros((sharding.shard_count(n_total, rank, world), out_len), dtype=torch.uint8, device=dev)

            def _decode(local):
                decode_into(local, d_loc_out)
                return d_loc_out

            def step():
                full = sharding.decode_sharded(d_all, n_total, FRAMEBITS, _decode)
                if rank == 0:
                    d_all_out.copy_(full)
    else:
        def step():
            decode_into(d_sym, d_out)

    # clock pre-conditioning (untimed, part of set-up like allocation and module load): same launches as a step
    if args.prewarm_ms > 0 and not args.stub:
        if args.mode != "shard" and dist:
            for _ in range(3):  # a fixed count keeps the ranks' collectives matched
                step()
        else:
            t_end = time.perf_counter() + args.prewarm_ms / 1e3
            while time.perf_counter() < t_end:
                for _ in range(8):
                    step()
                torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync()
    if dist:
        dist.barrier()
    sync()
    # HIP events on the launch stream bracket the whole timed region (one pair, not one per step: an event between
    # two launches costs the GPU a few microseconds of idle time per step)
    ev0 = ev1 = None
    if not args.stub:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if ev0 is not None:
        ev0.record()
    for _ in range(args.steps):
        step()
    if ev1 is not None:
        ev1.record()
    sync()
    if dist:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / max(1, args.steps) if ev0 is not None else dt / max(1, args.steps) * 1e3
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.mode != "shard" and dist and rank == 0 and args.stub:
        routing_ok = bool(torch.equal(d_all_out, d_all[:, :out_len]))  # every frame came back to its own row

    result = None
    if rank == 0:
        total_bits = float(world) * n * FRAMEBITS * args.steps
        alg_bytes = n * (4 * (FRAMEBITS + TAIL) + out_len)  # 3192 B per FIC frame
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        sharding_note = {"shard": "independent shards per rank, no data-path collective",
                         "scatter": "rank 0 owns all frames: chunked, overlapped RCCL send/recv pipeline "
                                    "(%d frames per rank per chunk) inside the step" % args.chunk_frames,
                         "scatter-plain": "rank 0 owns all frames: one round-robin scatter + one gather inside the step"}
        result = {
            "metric": "decoded Mbit/s per GPU on batched 768-bit DAB FIC frames; bit-exact vs AVX2 ref",
            "value": round(total_bits / dt / 1e6, 1),
            "unit": "Mbit/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "batch=%d FIC frames (768 bit, 3096 soft symbols u8) per GPU, "
                                   "resident in HBM; %s" % (n, "Eb/N0=3 dB reference-style noise" if args.input == "noisy"
                                                             else "UNIFORM RANDOM BYTES (not the headline workload)"),
                       "input": args.input,
                       "frames_per_gpu": n, "framebits": FRAMEBITS, "kernel": args.kernel, "renorm_ge": int(bool(args.renorm_ge)),
                       "renorm_comparator": ">= 150 (reference MASM decoders, decon_avx2.asm:97,114)" if args.renorm_ge else
                                            "> 150 (reference C decoders, deconvolve.cpp:408)",
                       "prewarm_ms": args.prewarm_ms,
                       "sharding": sharding_note[args.mode],
                       "launch": "spawned by bench.py" if os.environ.get("VIT_BENCH_SPAWNED") else
                                 ("external launcher" if spawned else "single process")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": alg_bytes},
        }
        if dist:
            # the N > 1 record describes itself: what moved over xGMI inside the timed region, and what did not
            result["mode"] = args.mode
            result["backend"] = dist.get_backend()
            result["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
            result["xgmi_bytes_in_timed_region"] = 0 if args.mode == "shard" else "symbols out + decoded bytes back, every step"
        if args.stub:
            result["stub"] = True
            result["data"] = "stub (CPU rehearsal of the launcher; value is NOT a measurement)"
            result["roofline"] = None
            if routing_ok is not None:
                result["routing_ok"] = routing_ok
        if args.mode != "shard" and result.get("roofline"):
            result["roofline"]["note"] = "kernel_ms here spans scatter+decode+gather; see shard mode for the kernel"
        if world == 1 and not args.no_cpu and not args.stub:
            O = _vitpkg.load_oracle()  # checker + timed CPU baseline only
            in_shard_mode = args.mode == "shard" or not dist
            sym_host = (d_sym if in_shard_mode else d_all).cpu().numpy()
            base, ref = cpu_baseline(O, sym_host, FRAMEBITS)
            if args.renorm_ge:  # the timed port implements `> 150`; the checker for this mode is the scalar ge oracle
                ref = O.decode_batch(FRAMEBITS, sym_host, nthreads=len(os.sched_getaffinity(0)), ge=True)
            got = (d_out if in_shard_mode else d_all_out).cpu().numpy()
            bad = int((got != ref).any(axis=1).sum())
            result["cpu_baseline"] = base
            result["parity"] = {"frames_checked": n, "frames_differing": bad, "bit_exact": bad == 0,
                                "checker": "this repo's own port of the reference's integer specification (oracle/, "
                                           "'parity unpinned' beyond SURVEY 8c's KATs - DESIGN.md (c)), not the reference binary"}
            result["speedup_vs_cpu_1thread"] = round(result["value"] / base["value"], 1)
            if bad:
                result["value"] = 0.0  # a fast kernel with wrong results is not a result
        if world == 1 and not args.stub and args.mode == "shard" and not args.no_pipelined:
            try:
                result["pipelined"] = pipelined_leg(V, d_sym, d_out)
                if result.get("parity") and not result["parity"]["bit_exact"]:
                    result["pipelined"]["value"] = 0.0
            except Exception as e:  # the headline line must not depend on this leg
                result["pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.stub and args.mode == "shard" and not args.no_sensitivity and args.input == "noisy":
            try:
                sens = input_sensitivity(V, None if args.no_cpu else _vitpkg.load_oracle(), dev, n, 1234 + rank, args.renorm_ge)
                for rec in sens.values():
                    rec["vs_headline"] = round(rec["value"] / result["value"] - 1.0, 4) if result["value"] else None
                sens["what"] = ("the same batch shape on other input families, unit Mbit/s; vs_headline = relative to `value` "
                                "(Eb/N0 = 3 dB); random_bytes = no signal at all, the speculative traceback's worst case")
                result["input_sensitivity"] = sens
            except Exception as e:  # the headline line must not depend on this leg
                result["input_sensitivity"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.stub and not args.no_rs and args.mode == "shard":
            try:
                result["second_stage"] = second_stage(V, dev)
            except Exception as e:  # the headline line must not depend on this leg
                result["second_stage"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.stub:
            _attach_cached_counters(result)
    if dist and world > 1 and args.mode == "shard" and not args.no_scatter_leg:
        # The extra leg is collective: every rank runs it.  It must never cost the run its headline line, so a
        # watchdog ends ALL ranks (rank 0 printing the line with the error in it, exit code 3) if it does not come back.
        import threading
        from importlib import import_module

        def bail():
            # a hung collective is a finding, not a success: the shard line is still printed (with the error in it), the
            # exit code of every rank is non-zero, and the launcher (spawn_ranks / torchrun) passes it on
            if rank == 0:
                result["scatter"] = {"error": "the scatter leg did not finish within %d s (RCCL send/recv pipeline hung or "
                                              "far slower than predicted); the shard measurement above it completed before "
                                              "the leg started; exit code 3" % args.scatter_timeout}
                print(json.dumps(result), flush=True)
            os._exit(3)

        dog = threading.Timer(args.scatter_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            scatter = scatter_leg(args, dist, import_module("viterbi_dll_amd.sharding"), rank, world, dev, d_sym, d_out,
                                  decode_into, sync, n * args.steps / dt)
        except Exception as e:  # noqa: BLE001 -- reported in the line, the shard result stands
            scatter = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None
        dog.cancel()
        if rank == 0 and scatter is not None:
            if "value" in scatter:
                one_gpu = result["value"] / world
                scatter["speedup_vs_1gpu_measured"] = round(scatter["value"] / one_gpu, 2) if one_gpu > 0 else None
                scatter["reading"] = ("`value` above is per-GPU ingestion (every rank owns its shard, nothing crosses xGMI: ~N x); "
                                      "this leg is ONE root feeding N-1 peers over one xGMI link each: bounded near 2 x whatever N")
            result["scatter"] = scatter
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


def main_multi(args):
    """ONE process, --gpus devices, the C-ABI pipeline: the whole stream (frames x gpus) lives on GPU 0."""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    V = _vitpkg.load_package()
    assert V.device_count() >= args.gpus, "libviterbi.so sees %d gfx950 device(s)" % V.device_count()
    V.initialize()
    V.set_kernel(args.kernel)
    n_total = args.frames * args.gpus
    out_len = (FRAMEBITS + 7) // 8
    d_all = make_frames(n_total, FRAMEBITS, seed=99, device=dev)
    d_all_out = torch.zeros((n_total, out_len), dtype=torch.uint8, device=dev)
    devices = list(range(args.gpus))
    flags = V.MULTI_LOOPBACK if args.loopback else 0
    rootf = -1 if args.root_frames is None else args.root_frames

    def step():
        V.decode_stream_multi(d_all, d_all_out, FRAMEBITS, n_total, devices, args.chunk_frames, rootf, flags)

    sys.stdout.flush()
    saved_stdout = os.dup(1)  # RCCL prints a version banner to stdout when its first communicator comes up
    os.dup2(2, 1)
    try:
        for _ in range(max(1, args.warmup)):
            step()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # synchronous
    dt = time.perf_counter() - t0
    ref = torch.zeros_like(d_all_out)
    V.decode_batch_dev(d_all, ref, FRAMEBITS, n_total)
    torch.cuda.synchronize()
    same = bool(torch.equal(ref, d_all_out))
    result = {
        "metric": "decoded Mbit/s per GPU on batched 768-bit DAB FIC frames; bit-exact vs AVX2 ref",
        "value": round(n_total * FRAMEBITS * args.steps / dt / 1e6, 1) if same else 0.0, "unit": "Mbit/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong-per-stream", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "stream of %d FIC frames resident on GPU 0, decoded by %d GPU(s) through "
                               "vit_decode_stream_multi (RCCL send/recv pipeline)" % (n_total, args.gpus),
                   "chunk_frames": args.chunk_frames, "root_frames": rootf, "loopback": bool(args.loopback),
                   "launch": "single process, C ABI"},
        "roofline": None, "multi_matches_single_launch": same,
    }
    print(json.dumps(result), flush=True)
    return result


def _attach_cached_counters(result):
    """PMC-derived figures cannot be collected inside a timed run: they come from the committed rocprofv3 passes
    of this same command (tools/collect_profiles.sh -> profiles/pmc_traffic.json) and are labelled as such."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return
    r = result["roofline"]
    # the counters belong to ONE configuration (the file records it): attach them only to a run of that configuration
    cfg, mine = pmc.get("config") or {}, result["config"]
    ran = {"frames_per_gpu": mine.get("frames_per_gpu"), "kernel": mine.get("kernel"), "mode": result.get("mode", "shard"),
           "renorm_ge": mine.get("renorm_ge", 0), "input": mine.get("input", "noisy")}
    want = {"frames_per_gpu": cfg.get("frames_per_gpu", 65536), "kernel": cfg.get("kernel", 0), "mode": cfg.get("mode", "shard"),
            "renorm_ge": cfg.get("renorm_ge", 0), "input": cfg.get("input", "noisy")}
    if ran != want:
        r["traffic_source"] = "profiles/pmc_traffic.json was measured for %s, this run is %s: counters not attached" % (want, ran)
        return
    r["traffic"] = pmc.get("hbm_bytes_per_launch")
    r["traffic_source"] = "cached from profiles/pmc_traffic.json (%s)" % pmc.get("source", "rocprofv3 --pmc passes")
    for k in ("valu_busy", "valu_insts_per_frame_step", "valu_insts_per_wave"):
        if k in pmc:
            r[k] = pmc[k]
    if "valu_busy" in pmc:
        r["binding_resource"] = "VALU issue (see roofline_valu); the HBM frac is reported as the north star asks"
    if "valu_busy" in pmc and "valu_insts_per_frame_step" in pmc:
        # the binding roofline: vector-instruction issue.  `core` = what the formulation cannot do without per frame-step:
        # 4 v_pk_add_u16 + 2 v_pk_min_u16 + 2 v_pk_sub_u16 (decisions) + 2 v_bfi (history) + 1 v_sub (63 - M) = 11 wave
        # instructions per trellis step of FOUR frames; issued = SQ_INSTS_VALU per wave / (4 x 774).
        core, issued, busy = 11.0 / 4.0, float(pmc["valu_insts_per_frame_step"]), float(pmc["valu_busy"])
        result["roofline_valu"] = {
            "bound": "valu_issue", "unit": "wave instructions per frame-step",
            "core": core, "issued": issued, "valu_busy": busy,
            "useful_frac": round(core / issued * busy, 4),
            "overhead_frac_of_issued": round(1.0 - core / issued, 4),
            "source": r["traffic_source"],
            "what": "useful_frac = core / issued x valu_busy: the share of the chip's vector issue slots spent on the "
                    "add-compare-select + history core; the rest is renormalisation, lane exchange, pre-pass, traceback, set-up and idle slots"}
    ss = result.get("second_stage")
    if ss and ss.get("roofline") and pmc.get("second_stage"):
        ss["roofline"]["traffic"] = pmc["second_stage"].get("hbm_bytes_per_launch")
        ss["roofline"]["traffic_source"] = r["traffic_source"]


if __name__ == "__main__":
    main()
