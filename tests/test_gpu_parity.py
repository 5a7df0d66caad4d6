"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact is the bar: every decoded byte and every RS return value / output byte.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import os

# 1 = wave-per-frame kernel, 2 = packed 4-frames-per-wave kernel, 3 = latency kernel (one frame per wave, DPP)
KERNELS = [int(k) for k in os.environ.get("VIT_TEST_KERNELS", "1,2,3").split(",")]


def _gpu_decode(V, torch, sym, framebits, kernel):
    n = sym.shape[0]
    old = V.set_kernel(kernel)
    try:
        d_sym = torch.from_numpy(np.ascontiguousarray(sym)).cuda()
        d_out = torch.full((n, (framebits + 7) // 8), 0xEE, dtype=torch.uint8, device="cuda")
        V.decode_batch_dev(d_sym, d_out, framebits, n)
        torch.cuda.synchronize()
        return d_out.cpu().numpy()
    finally:
        V.set_kernel(old)


def _mixed_input(O, n, framebits, seed):
    """half reference-style noisy frames (Eb/N0 3 dB), half adversarial uniform bytes"""
    a = O.noisy_frames(n - n // 2, framebits, seed=seed)
    b = O.uniform_symbols((n // 2) * O.sym_len(framebits), seed=seed + 1000).reshape(n // 2, -1)
    return np.concatenate([a, b])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("framebits", [768, 288, 1536, 2304, 3072, 8, 16, 96, 2, 10, 770, 778, 780, 1542, 3070])
def test_decode_parity_uniform_length(V, O, torch_cuda, framebits, kernel):
    n = 203 if framebits <= 3072 else 37  # not a multiple of 4: ragged last group
    sym = _mixed_input(O, n, framebits, seed=framebits + 1)
    want = O.decode_batch(framebits, sym, nthreads=8)
    got = _gpu_decode(V, torch_cuda, sym, framebits, kernel)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("framebits", [6912, 9216, 9214, 5002])
def test_decode_parity_long_frames(V, O, torch_cuda, framebits, kernel):
    n = 13
    sym = _mixed_input(O, n, framebits, seed=framebits)
    want = O.decode_batch(framebits, sym, nthreads=8)
    old = V.set_kernel(kernel)
    try:
        supported = True
        try:
            got = _gpu_decode(V, torch_cuda, sym, framebits, kernel)
        except V.ViterbiError:
            supported = False
    finally:
        V.set_kernel(old)
    if not supported:
        assert kernel == 2, "the wave kernel must handle every length"
        pytest.skip("packed kernel does not take %d-bit frames (auto falls back to the wave kernel)" % framebits)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("framebits", [784, 1008, 1040, 3072, 6912, 9216])
def test_long_frames_in_flight_parts(V, O, torch_cuda, framebits):
    """the long-frame kernel's in-flight parts (csrc/vit_pk.hip, round 4: a group of four equally long frames is traced back
    256 steps at a time WHILE the add-compare-select runs ahead, from a speculative top position; the chain of recorded
    positions is checked after the forward pass and a part that fails comes back from the write-only spill) on the input
    families that take each of its paths: Eb/N0 3 dB (about 1 % of the wave-parts fail their check), 2 dB (7 %), 1 dB (most
    waves give up tracing in flight after their first part), 0 dB / uniform random bytes / hard decisions (all give up) -
    tests/tools/spec_stats.py counts them on a -DVIT_DIAG_SPEC build (profiles/r04_spec_stats.jsonl)"""
    n = 192 if framebits <= 3072 else 96
    fams = [O.noisy_frames(n, framebits, seed=framebits + 11),
            O.noisy_frames(n, framebits, seed=framebits + 12, ebn0_db=2.0),
            O.noisy_frames(n, framebits, seed=framebits + 13, ebn0_db=1.0),
            O.noisy_frames(n // 2, framebits, seed=framebits + 14, ebn0_db=0.0),
            O.uniform_symbols((n // 2) * O.sym_len(framebits), seed=framebits + 15).reshape(n // 2, -1),
            O.hard_flipped_frames(n // 2, framebits, flip=0.1, seed=framebits + 16)]
    sym = np.concatenate(fams)
    want = O.decode_batch(framebits, sym, nthreads=8)
    got = _gpu_decode(V, torch_cuda, sym, framebits, 2)
    bad = np.flatnonzero((got != want).any(axis=1))
    assert bad.size == 0, "frames %s differ" % bad[:8]


def test_long_frames_in_flight_parts_in_a_descriptor_table(V, O, torch_cuda):
    """the same through the variable-length entry: groups of four equal lengths take the in-flight form, groups that
    straddle two lengths (and lengths that are not a multiple of 16) the general form - both inside one launch, every
    frame compared"""
    import torch
    rng = np.random.default_rng(44)
    lens = [6912] * 10 + [4608] * 9 + [3072] * 7 + [1000] * 5 + [784] * 6 + [782] * 3 + [9216] * 2 + [288] * 5
    rng.shuffle(lens)
    syms, wants, descs, so, oo = [], [], [], 0, 0
    for i, fb in enumerate(lens):
        s = O.noisy_frames(1, fb, seed=1000 + i, ebn0_db=(3.0 if i % 3 else 2.0))[0]
        syms.append(s)
        wants.append(O.decode_batch(fb, s[None, :])[0])
        descs.append((so, oo, fb, 0))
        so += s.size
        oo += (fb + 7) // 8
    d_sym = torch.from_numpy(np.concatenate(syms)).cuda()
    d_out = torch.full((oo,), 0xEE, dtype=torch.uint8, device="cuda")
    desc = np.array(descs, dtype=V.DESC_DTYPE)
    old = V.set_kernel(2)
    try:
        V.decode_varlen_dev(d_sym, d_out, torch.from_numpy(desc.view(np.uint8)).cuda(), len(lens), max(lens))
        torch.cuda.synchronize()
    finally:
        V.set_kernel(old)
    got = d_out.cpu().numpy()
    for (so, oo, fb, _), w in zip(descs, wants):
        assert np.array_equal(got[oo:oo + (fb + 7) // 8], w), fb


def test_fast_traceback_form_every_multiple_of_16(V, O, torch_cuda):
    """the straight-line traceback form (csrc/vit_pk.hip traceback_part16: waves of four equally long frames, a multiple
    of 16 bits) over EVERY such length up to 1600 bits - all register/LDS window shapes of the single-segment kernel
    (R = 0..32, partial last window shift) and the first spilled groups of the long-frame kernel - plus long ones; 12
    frames = three uniform waves; reference-style noise, uniform bytes and hard decisions (re-trace passes)"""
    lengths = list(range(16, 1601, 16)) + [2048, 3056, 4096, 4112, 5008, 6912, 9200, 9216]
    for fb in lengths:
        sym = np.concatenate([_mixed_input(O, 8, fb, seed=fb), O.hard_random_symbols(4, fb, seed=fb + 5)])
        want = O.decode_batch(fb, sym, nthreads=8)
        got = _gpu_decode(V, torch_cuda, sym, fb, 2)
        assert np.array_equal(got, want), "framebits=%d" % fb


@pytest.mark.parametrize("extra_groups", [0, 3, 4097])
def test_launch_shapes_around_one_round_of_waves(V, O, torch_cuda, extra_groups):
    """Launches of exactly / just over one round of resident workgroups: the single-segment kernel switches between its two
    instantiations (rotating issue priority for one round of waves), the persistent long-frame kernel hands every workgroup its
    first group statically and the later ones from the counter.  Every output byte of every frame is compared."""
    resident = 16 * torch_cuda.cuda.get_device_properties(0).multi_processor_count  # 10 KB of LDS per workgroup: 16 per CU
    for fb, seed in ((96, 41), (784, 42)):  # 96 bits: single-segment kernel; 784: the shortest frames of the long-frame kernel
        n = 4 * (resident + extra_groups) - (1 if extra_groups == 3 else 0)  # once with a ragged last group
        base = _mixed_input(O, 512, fb, seed=seed)
        reps = (n + 511) // 512
        sym = np.tile(base, (reps, 1))[:n]
        want = np.tile(O.decode_batch(fb, base, nthreads=8), (reps, 1))[:n]
        got = _gpu_decode(V, torch_cuda, sym, fb, 2)
        bad = int((got != want).any(axis=1).sum())
        assert got.shape == want.shape and bad == 0, "framebits %d, %d groups: %d frames differ" % (fb, (n + 3) // 4, bad)


def test_auto_kernel_handles_max_length(V, O, torch_cuda):
    framebits, n = 9216, 5
    sym = _mixed_input(O, n, framebits, seed=5)
    want = O.decode_batch(framebits, sym, nthreads=8)
    got = _gpu_decode(V, torch_cuda, sym, framebits, 0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kernel", KERNELS)
def test_saturation_and_renorm_stress(V, O, torch_cuda, kernel):
    """inputs built to drive metrics into the 255 clamp and the subs-63 floor"""
    framebits, n = 768, 64
    rng = np.random.default_rng(11)
    sl = O.sym_len(framebits)
    sym = np.empty((n, sl), np.uint8)
    sym[0::4] = 0
    sym[1::4] = 255
    sym[2::4] = rng.integers(0, 2, (n // 4, sl), dtype=np.uint8) * 255
    blk = rng.integers(0, 256, (n // 4, sl // 64 + 1), dtype=np.uint8)
    sym[3::4] = np.repeat(blk, 64, axis=1)[:, :sl]
    want = O.decode_batch(framebits, sym)
    got = _gpu_decode(V, torch_cuda, sym, framebits, kernel)
    assert np.array_equal(got, want)


def _hard_families(O, framebits, n, seed):
    """the two input families on which the reference's two renormalise comparators give different outputs"""
    return np.concatenate([O.hard_random_symbols(n, framebits, seed=seed),
                           O.hard_flipped_frames(n, framebits, flip=0.2, seed=seed + 1)])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("framebits,n", [(768, 120), (3072, 40), (6912, 16)])
def test_renorm_ge_mode(V, O, torch_cuda, kernel, framebits, n):
    """vit_set_renorm_ge(1): the `>= 150` renormalise test of the reference's MASM decoders (decon_avx2.asm:94-118,
    the shipped Rel_asm configuration) against the oracle's ge mode; mode 0 (`> 150`, deconvolve.cpp:407-412, Rel_cpp)
    against the gt oracle on the SAME frames -- hard-decision input, where the two modes really differ (asserted).
    The oracle's ge mode is restated from the asm text, which cannot be assembled here: parity unpinned."""
    sym = _hard_families(O, framebits, n, seed=framebits)
    want_gt = O.decode_batch(framebits, sym, nthreads=8)
    want_ge = O.decode_batch(framebits, sym, nthreads=8, ge=True)
    assert (want_gt != want_ge).any(axis=1).sum() >= 3, "the frame set does not separate the two comparators"
    assert V.set_renorm_ge(0) == 0  # the test session's mode (conftest.py)
    assert np.array_equal(_gpu_decode(V, torch_cuda, sym, framebits, kernel), want_gt)
    V.set_renorm_ge(1)
    try:
        got = _gpu_decode(V, torch_cuda, sym, framebits, kernel)
    finally:
        assert V.set_renorm_ge(0) == 1
    assert np.array_equal(got, want_ge)


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
def test_renorm_ge_mode_varlen_and_exports(V, O, torch_cuda, kernel):
    """ge mode through the variable-length entry point (sorted + split tables, long and short frames), the u32 ingest
    path and the drop-in deconvolve() export"""
    torch = torch_cuda
    rng = np.random.default_rng(150)
    fbs = [768] * 40 + [3072] * 10 + [6912] * 6 + [288] * 12 + [770, 2, 1536]
    rng.shuffle(fbs)
    fbs = [int(x) for x in fbs]
    desc, sym_bytes, out_bytes = V.make_descs(fbs)
    sym = np.empty(sym_bytes, np.uint8)
    want = {False: np.empty(out_bytes, np.uint8), True: np.empty(out_bytes, np.uint8)}
    for i, (fb, d) in enumerate(zip(fbs, desc)):
        so, oo = int(d["sym_offset"]), int(d["out_offset"])
        fr = (O.hard_flipped_frames(1, fb, flip=0.2, seed=1000 + i) if i & 1 else O.hard_random_symbols(1, fb, seed=1000 + i))
        sym[so:so + O.sym_len(fb)] = fr[0]
        for ge in (False, True):
            want[ge][oo:oo + (fb + 7) // 8] = O.decode_batch(fb, fr, ge=ge)[0]
    assert not np.array_equal(want[False], want[True])
    d_sym = torch.from_numpy(sym).cuda()
    d_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    old = V.set_kernel(kernel)
    try:
        for ge in (True, False):
            V.set_renorm_ge(ge)
            d_out = torch.zeros(out_bytes, dtype=torch.uint8, device="cuda")
            V.decode_varlen_dev(d_sym, d_out, d_desc, len(fbs), max(fbs))
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), want[ge]), "varlen ge=%s" % ge
        # u32 ingest + the single-frame export
        fb = 3072
        fr = O.hard_flipped_frames(24, fb, flip=0.2, seed=9)
        w_gt, w_ge = O.decode_batch(fb, fr), O.decode_batch(fb, fr, ge=True)
        pick = int(np.flatnonzero((w_gt != w_ge).any(axis=1))[0])
        d32 = torch.from_numpy(fr.astype(np.int32)).cuda()
        for ge, w in ((True, w_ge), (False, w_gt)):
            V.set_renorm_ge(ge)
            d_out = torch.zeros((24, fb // 8), dtype=torch.uint8, device="cuda")
            V.decode_batch_dev_u32(d32, d_out, fb, 24)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy(), w), "u32 ge=%s" % ge
            rc, one = V.deconvolve(fb, fr[pick].astype(np.uint32))
            assert rc == 0 and np.array_equal(one, w[pick]), "deconvolve ge=%s" % ge
            assert np.array_equal(one, O.deconvolve_u32(fb, fr[pick].astype(np.uint32), ge=ge))
    finally:
        V.set_renorm_ge(0)
        V.set_kernel(old)


@pytest.mark.parametrize("kernel", KERNELS)
def test_noise_free_roundtrip(V, O, torch_cuda, kernel):
    framebits, n = 768, 32
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, (n, framebits), dtype=np.uint8)
    sym = np.stack([O.encode(b) * 255 for b in bits]).astype(np.uint8)
    got = _gpu_decode(V, torch_cuda, sym, framebits, kernel)
    assert np.array_equal(np.unpackbits(got, axis=1), bits)


def test_deconvolve_export_reference_abi(V, O, torch_cuda):
    """the drop-in single-frame call: u32 symbols, host pointers, returns 0"""
    for framebits in (768, 3072, 8, 9216, 770):
        sym = O.uniform_symbols(O.sym_len(framebits), seed=framebits)
        s32 = sym.astype(np.uint32) | np.uint32(0xABCD0000 if framebits == 768 else 0)  # only the low byte counts
        rc, got = V.deconvolve(framebits, s32)
        want = O.deconvolve_u32(framebits, s32)
        assert rc == 0
        assert np.array_equal(got, want)
    # framebits == 0 returns 0 and touches nothing
    assert V.lib().deconvolve(0, None, 0, None) == 0
    # bad arguments -> 1, never a crash
    assert V.lib().deconvolve(768, None, 0, None) == 1
    assert V.lib().deconvolve(9218, None, 0, None) == 1


def test_deconvolve_is_thread_safe(V, O, torch_cuda):
    import threading
    framebits = 768
    syms = O.noisy_frames(8, framebits, seed=3)
    want = O.decode_batch(framebits, syms)
    errs = []

    def work(i):
        for _ in range(20):
            rc, got = V.deconvolve(framebits, syms[i].astype(np.uint32))
            if rc != 0 or not np.array_equal(got, want[i]):
                errs.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs


def _run_callers(V, lens, syms, want, ncalls):
    import threading
    errs = []

    def work(i):
        for _ in range(ncalls):
            rc, got = V.deconvolve(lens[i], syms[i])
            if rc != 0 or not np.array_equal(got, want[i]):
                errs.append(i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(lens))]
    [t.start() for t in th]
    [t.join() for t in th]
    return errs


def test_deconvolve_micro_batching(V, O, torch_cuda):
    """ingest stage: concurrent callers share launches through the slot ring; results and return codes unchanged.
    Mixed lengths in one batch (every workgroup its own length), every engagement threshold, one to six batches in
    flight, waiting callers spinning (CPU budget large) and sleeping behind the batch's polling member (budget 0)."""
    lens = [768, 768, 3072, 288, 1536, 768, 9216, 8, 770, 2, 6912, 768]
    syms = [O.uniform_symbols(O.sym_len(fb), seed=fb + i).astype(np.uint32) | np.uint32(0x5A000000) for i, fb in enumerate(lens)]
    want = [O.deconvolve_u32(fb, s) for fb, s in zip(lens, syms)]
    old = V.set_batch_window_us(200)
    old_spin = V.set_batch_spin_cpus(1024)
    old_depth = V.set_batch_depth(3)
    try:
        # min_callers 1: every call goes through the ring; 8: only while eight calls are in flight; 100: never
        for min_callers in (1, 2, 8, 100):
            old_min = V.set_batch_min_callers(min_callers)
            try:
                assert not _run_callers(V, lens, syms, want, 12), min_callers
            finally:
                V.set_batch_min_callers(old_min)
        V.set_batch_min_callers(1)
        for depth, spin in ((1, 1024), (6, 1024), (1, 0), (2, 0), (3, 0), (3, 4)):
            V.set_batch_depth(depth)
            V.set_batch_spin_cpus(spin)
            assert not _run_callers(V, lens, syms, want, 12), (depth, spin)
        rc, got = V.deconvolve(770, syms[8])  # alone: a batch of one, launched at once
        assert rc == 0 and np.array_equal(got, want[8])
        V.set_batch_window_us(0)  # stage off: the direct path
        rc, got = V.deconvolve(770, syms[8])
        assert rc == 0 and np.array_equal(got, want[8])
    finally:
        V.set_batch_min_callers(1)
        V.set_batch_depth(old_depth)
        V.set_batch_spin_cpus(old_spin)
        V.set_batch_window_us(old)


def test_deconvolve_ring_many_callers_and_comparator(V, O, torch_cuda):
    """more concurrent callers than the ring has slots (128): the surplus takes the direct path, nobody fails or waits
    for a slot; and the `>= 150` comparator through the ring (hard-decision frames on which the two comparators differ)"""
    n = 150
    lens = [768] * n
    noisy = O.noisy_frames(n, 768, seed=11)
    syms = [noisy[i].astype(np.uint32) for i in range(n)]
    want = list(O.decode_batch(768, noisy))
    old = V.set_batch_window_us(100)
    old_spin = V.set_batch_spin_cpus(0)
    try:
        assert not _run_callers(V, lens, syms, want, 4)
        V.set_batch_spin_cpus(1024)
        assert not _run_callers(V, lens[:40], syms[:40], want[:40], 6)
        # comparator modes: 16 hard-decision frames, 3072 bits
        rng = np.random.default_rng(77)
        hard = [(rng.integers(0, 2, O.sym_len(3072)) * 255).astype(np.uint32) for _ in range(16)]
        w_gt = [O.deconvolve_u32(3072, h) for h in hard]
        w_ge = [O.deconvolve_u32(3072, h, ge=True) for h in hard]
        assert any(not np.array_equal(a, b) for a, b in zip(w_gt, w_ge))
        for ge, w in ((1, w_ge), (0, w_gt)):
            old_ge = V.set_renorm_ge(ge)
            try:
                assert not _run_callers(V, [3072] * 16, hard, w, 3), ge
            finally:
                V.set_renorm_ge(old_ge)
    finally:
        V.set_batch_spin_cpus(old_spin)
        V.set_batch_window_us(old)


@pytest.mark.parametrize("framebits,n", [(768, 50), (3072, 9), (24, 5), (6912, 40)])
def test_u32_ingest_path(V, O, torch_cuda, framebits, n):
    """the reference ABI's u32-per-symbol format resident on the device (only the low byte counts,
    deconvolve.cpp:158-165): read in place by the packed kernels (short and long frames), narrowed by the
    ingest kernel for the wave kernel and for a buffer that is not 16-byte aligned"""
    torch = torch_cuda
    sym = _mixed_input(O, n, framebits, seed=77)
    want = O.decode_batch(framebits, sym)
    junk = np.random.default_rng(5).integers(0, 1 << 24, sym.shape, dtype=np.int64) << 8  # upper bytes must be ignored
    host32 = ((sym.astype(np.int64) | junk) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    flat = torch.zeros(host32.size + 4, dtype=torch.int32, device="cuda")
    for kernel, shift in ((0, 0), (0, 1), (1, 0), (2, 0)):
        d32 = flat[shift:shift + host32.size]
        d32.copy_(torch.from_numpy(host32.reshape(-1)))
        d_out = torch.zeros((n, framebits // 8), dtype=torch.uint8, device="cuda")
        old = V.set_kernel(kernel)
        try:
            V.decode_batch_dev_u32(d32, d_out, framebits, n)
            torch.cuda.synchronize()
        finally:
            V.set_kernel(old)
        assert np.array_equal(d_out.cpu().numpy(), want), (kernel, shift)


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
def test_varlen_batch(V, O, torch_cuda, kernel):
    """BASELINE config 3 in small: framebits = 96*m, m in 3..72, descriptor table (+ lengths that are
    not multiples of 8: a partial last byte per frame)"""
    torch = torch_cuda
    rng = np.random.default_rng(1)
    fbs = (96 * rng.integers(3, 73, 97)).tolist() + [8, 9216, 2, 770, 4098, 9214, 30]  # any even length
    desc, sym_bytes, out_bytes = V.make_descs(fbs)
    sym = O.uniform_symbols(sym_bytes, seed=4)
    want = np.concatenate([O.decode_batch(fb, sym[int(d["sym_offset"]):int(d["sym_offset"]) + O.sym_len(fb)])[0]
                           for fb, d in zip(fbs, desc)])
    if kernel == 2:
        desc = V.sort_descs(desc.copy())  # the order of the table must not change any output byte
    old = V.set_kernel(kernel)
    try:
        d_sym = torch.from_numpy(sym).cuda()
        d_out = torch.zeros(out_bytes, dtype=torch.uint8, device="cuda")
        d_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
        V.decode_varlen_dev(d_sym, d_out, d_desc, len(fbs), max(fbs))
        torch.cuda.synchronize()
    finally:
        V.set_kernel(old)
    assert np.array_equal(d_out.cpu().numpy(), want)


def test_random_lengths_and_batch_sizes(V, O, torch_cuda):
    """sweep: every multiple of 8 around the segment / register-block boundaries plus random ones,
    with batch sizes that leave ragged groups; uniform-length entry point, auto kernel selection"""
    rng = np.random.default_rng(2025)
    edges = [8, 16, 24, 104, 112, 120, 248, 256, 264, 272, 504, 512, 520, 760, 768, 776, 784, 792, 1544, 1552, 1560,
             2336, 3128, 3136, 4096, 4104, 7840, 9208, 9216]
    lengths = sorted(set(edges + (8 * rng.integers(1, 1153, 12)).tolist() + (2 * rng.integers(1, 4609, 12)).tolist()))
    for fb in lengths:
        n = int(rng.integers(1, 10))
        sym = _mixed_input(O, n, fb, seed=fb) if n > 1 else O.noisy_frames(1, fb, seed=fb)
        want = O.decode_batch(fb, sym, nthreads=8)
        got = _gpu_decode(V, torch_cuda, sym, fb, 0)
        assert np.array_equal(got, want), "framebits=%d n=%d" % (fb, n)


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
def test_varlen_device_sort_mixed_and_invalid(V, O, torch_cuda, kernel):
    """the launcher length-sorts a copy of the table on the device (csrc/vit_sort.hip): a few thousand mixed
    lengths incl. long (spilled) frames, many equal keys, and descriptors the launch was not sized for --
    every valid frame bit-exact, invalid ones (too long, odd length, symbols not dword aligned) untouched,
    the caller's table unmodified; for the automatic choice and for both kernels"""
    torch = torch_cuda
    rng = np.random.default_rng(77)
    fbs = (8 * rng.integers(1, 40, 2500)).tolist() + [768] * 700 + (96 * rng.integers(3, 73, 60)).tolist() + [9216, 8]
    rng.shuffle(fbs)
    fbs = [int(x) for x in fbs]
    desc, sym_bytes, out_bytes = V.make_descs(fbs)
    bad_long, bad_odd, bad_align = 5, 1234, 2001
    sym = O.uniform_symbols(sym_bytes, seed=11)
    want = np.full(out_bytes, 0x5A, np.uint8)
    for i, (fb, d) in enumerate(zip(fbs, desc)):
        so, oo = int(d["sym_offset"]), int(d["out_offset"])
        if i not in (bad_long, bad_odd, bad_align):
            want[oo:oo + fb // 8] = O.decode_batch(fb, sym[so:so + O.sym_len(fb)])[0]
    desc["framebits"][bad_long] = 9216 + 8   # beyond max_framebits
    desc["framebits"][bad_odd] = fbs[bad_odd] + 1  # odd: not a valid frame length
    desc["sym_offset"][bad_align] += 2  # the kernels load one dword per trellis step
    d_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
    d_out = torch.full((out_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    old = V.set_kernel(kernel)
    try:
        V.decode_varlen_dev(torch.from_numpy(sym).cuda(), d_out, d_desc, len(fbs), 9216)
        torch.cuda.synchronize()
    finally:
        V.set_kernel(old)
    got = d_out.cpu().numpy()
    if not np.array_equal(got, want):
        bad = [(i, fbs[i]) for i, d in enumerate(desc)
               if not np.array_equal(got[int(d["out_offset"]):int(d["out_offset"]) + (fbs[i] + 7) // 8],
                                     want[int(d["out_offset"]):int(d["out_offset"]) + (fbs[i] + 7) // 8])]
        raise AssertionError("frames (index, framebits) that differ: %s" % bad[:12])
    assert np.array_equal(d_desc.cpu().numpy(), desc.view(np.uint8))
    if kernel:
        return
    # and a short-frame-only table (single-segment kernel behind the sort)
    fbs2 = [int(x) for x in 8 * rng.integers(1, 97, 500)]
    desc2, sb2, ob2 = V.make_descs(fbs2)
    sym2 = O.uniform_symbols(sb2, seed=12)
    want2 = np.concatenate([O.decode_batch(fb, sym2[int(d["sym_offset"]):int(d["sym_offset"]) + O.sym_len(fb)])[0]
                            for fb, d in zip(fbs2, desc2)])
    d_out2 = torch.zeros(ob2, dtype=torch.uint8, device="cuda")
    V.decode_varlen_dev(torch.from_numpy(sym2).cuda(), d_out2, torch.from_numpy(desc2.view(np.uint8)).cuda(), len(fbs2),
                        max(fbs2))
    torch.cuda.synchronize()
    assert np.array_equal(d_out2.cpu().numpy(), want2)


@pytest.mark.parametrize("kernel", [0, 2])
def test_split_table_handover_inside_a_sort_bin(V, O, torch_cuda, kernel):
    """A length-sorted table that is split between the two packed kernels (fewer than 1/8 of the frames long): the sort
    orders by framebits/8 only, so bin 98 holds 778-bit frames (single-segment kernel) next to 780/782/784-bit ones
    (long-frame kernel) in input order.  More all-778 groups than the long-frame kernel has workgroups, THEN the
    780..784-bit frames: every one of them must still be decoded (round-2 advisor finding: each persistent workgroup
    left at its first all-778 group).  Descriptors share a few distinct symbol blocks, outputs are all distinct."""
    torch = torch_cuda
    # the sort places a workgroup's descriptors in the order the workgroups reach the bin's cursor: the 780..784-bit
    # frames sit at the very end of the input so that they land behind (nearly) all of the 100000 778-bit ones
    lens = [768] * 400000 + [778] * 100000 + [768] * 400000 + [3072] * 4 + [780, 782, 784] * 16
    distinct = {}
    sym_parts, pos = [], 0
    for fb in (778, 780, 782, 784, 768, 3072):
        fr = _mixed_input(O, 8, fb, seed=fb)
        distinct[fb] = (pos, O.decode_batch(fb, fr))
        sym_parts.append(fr.reshape(-1))
        pos += fr.size
    sym = np.concatenate(sym_parts)
    desc = np.zeros(len(lens), V.DESC_DTYPE)
    fbs = np.asarray(lens, np.int64)
    which = np.arange(len(lens)) % 8
    desc["framebits"] = fbs
    desc["sym_offset"] = [distinct[fb][0] for fb in lens] + which * 4 * (fbs + 6)
    osz = (fbs + 7) // 8
    desc["out_offset"] = np.concatenate(([0], np.cumsum(osz)[:-1]))
    out_bytes = int(osz.sum())
    d_out = torch.full((out_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
    old = V.set_kernel(kernel)
    try:
        V.decode_varlen_dev(torch.from_numpy(sym).cuda(), d_out, torch.from_numpy(desc.view(np.uint8)).cuda(), len(lens), 3072)
        torch.cuda.synchronize()
    finally:
        V.set_kernel(old)
    got = d_out.cpu().numpy()
    bad = []
    for fb in (780, 782, 784, 778, 3072, 768):
        idx = np.flatnonzero(fbs == fb)
        nb = (fb + 7) // 8
        rows = got[(desc["out_offset"][idx][:, None] + np.arange(nb)[None, :]).astype(np.int64)]
        ok = (rows == distinct[fb][1][which[idx]]).all(axis=1)
        if not ok.all():
            bad.append((fb, int((~ok).sum()), int(idx[np.flatnonzero(~ok)[0]])))
    assert not bad, "(framebits, frames that differ, first index): %s" % bad


@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
def test_varlen_checked_skips_descriptors_outside_the_buffers(V, O, torch_cuda, kernel):
    """vit_decode_varlen_dev_checked: a descriptor whose symbols or output bytes reach outside the two buffers (far
    outside, or by one byte) is skipped on the device like the other invalid ones - nothing read, nothing written;
    the rest of the table decodes bit-exact.  Table sizes below and above the sort threshold (16)."""
    torch = torch_cuda
    rng = np.random.default_rng(21)
    for nfr in (9, 700):
        fbs = [int(x) for x in 8 * rng.integers(1, 130, nfr)] + [3072, 768]
        desc, sym_bytes, out_bytes = V.make_descs(fbs)
        sym = O.uniform_symbols(sym_bytes, seed=nfr)
        want = np.full(out_bytes, 0x5A, np.uint8)
        bad = {1: ("sym_offset", 1 << 40), 3: ("out_offset", 1 << 41),
               5: ("sym_offset", sym_bytes - O.sym_len(fbs[5]) + 4),   # the last dword lies outside
               6: ("out_offset", out_bytes - fbs[6] // 8 + 1),         # the last byte lies outside
               len(fbs) - 1: ("sym_offset", (1 << 64) - 4)}            # offset + size wraps around
        for i, (fb, d) in enumerate(zip(fbs, desc)):
            so, oo = int(d["sym_offset"]), int(d["out_offset"])
            if i not in bad:
                want[oo:oo + fb // 8] = O.decode_batch(fb, sym[so:so + O.sym_len(fb)])[0]
        for i, (field, val) in bad.items():
            desc[field][i] = val
        # the last descriptor of the ORIGINAL layout ends exactly at the end of both buffers: it must be decoded
        assert len(fbs) - 2 not in bad
        d_desc = torch.from_numpy(desc.view(np.uint8)).cuda()
        d_out = torch.full((out_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
        old = V.set_kernel(kernel)
        try:
            V.decode_varlen_dev_checked(torch.from_numpy(sym).cuda(), d_out, d_desc, len(fbs), 3072)
            torch.cuda.synchronize()
        finally:
            V.set_kernel(old)
        assert np.array_equal(d_out.cpu().numpy(), want), (kernel, nfr)
        assert np.array_equal(d_desc.cpu().numpy(), desc.view(np.uint8))  # the caller's table is not modified


def test_long_frames_from_threads_and_streams(V, O, torch_cuda):
    """the long-frame kernel's spill/sort scratch is per calling thread and its reuse is ordered by an
    event: two threads, each alternating between two streams with different frame lengths, back to back"""
    import threading
    torch = torch_cuda
    cases = [(3072, 37), (1536, 61), (9216, 9), (2304, 33)]
    data = []
    for fb, n in cases:
        sym = _mixed_input(O, n, fb, seed=fb)
        data.append((fb, n, torch.from_numpy(sym).cuda(), O.decode_batch(fb, sym, nthreads=8)))
    errs = []

    def work(tid):
        try:
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            outs = []
            for rep in range(6):
                fb, n, d_sym, want = data[(tid + rep) % len(data)]
                st = streams[rep & 1]
                with torch.cuda.stream(st):
                    d_out = torch.zeros((n, fb // 8), dtype=torch.uint8, device="cuda")
                    V.decode_batch_dev(d_sym, d_out, fb, n, stream=st.cuda_stream)
                outs.append((d_out, want, fb))
            torch.cuda.synchronize()
            for d_out, want, fb in outs:
                if not np.array_equal(d_out.cpu().numpy(), want):
                    errs.append((tid, fb))
        except Exception as e:  # noqa: BLE001
            errs.append((tid, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_varlen_descriptor_longer_than_declared_is_skipped(V, O, torch_cuda):
    torch = torch_cuda
    fbs = [768, 1536, 768, 768, 288]
    desc, sym_bytes, out_bytes = V.make_descs(fbs)
    sym = O.uniform_symbols(sym_bytes, seed=8)
    for kernel in (1, 2):
        old = V.set_kernel(kernel)
        try:
            d_out = torch.full((out_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
            V.decode_varlen_dev(torch.from_numpy(sym).cuda(), d_out, torch.from_numpy(desc.view(np.uint8)).cuda(),
                                len(fbs), 768)  # the 1536-bit frame exceeds the declared maximum
            torch.cuda.synchronize()
        finally:
            V.set_kernel(old)
        got = d_out.cpu().numpy()
        for fb, d in zip(fbs, desc):
            so, oo = int(d["sym_offset"]), int(d["out_offset"])
            if fb > 768:
                assert (got[oo:oo + fb // 8] == 0x5A).all()
            else:
                assert np.array_equal(got[oo:oo + fb // 8], O.decode_batch(fb, sym[so:so + O.sym_len(fb)])[0])


# ---- RS(120,110) -----------------------------------------------------------------

def _rs_superframes(O, nsf, rsdims, seed, max_err=7):
    """valid codewords column-wise with 0..max_err injected symbol errors per column"""
    rng = np.random.default_rng(seed)
    p = np.empty((nsf, 120, rsdims), np.uint8)
    for s in range(nsf):
        for j in range(rsdims):
            cw = O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8))
            ne = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 5, 6, max_err]))
            if s % 3 == 0:
                ne = min(ne, 5)  # superframes without failures
            pos = rng.choice(120, ne, replace=False)
            cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
            p[s, :, j] = cw
    return p.reshape(nsf, 120 * rsdims)


@pytest.mark.parametrize("rsdims", [24, 12, 4, 1, 48, 7])
def test_rs_batch_parity(V, O, torch_cuda, rsdims):
    torch = torch_cuda
    nsf = 41
    p = _rs_superframes(O, nsf, rsdims, seed=rsdims)
    init = np.full((nsf, 110 * rsdims), 0xA5, np.uint8)  # sentinel: untouched columns must keep it
    ret_ref, out_ref = O.rs_check_batch(p, rsdims, out_init=init)
    assert (ret_ref == -1).any() and (ret_ref > 0).any()
    d_p = torch.from_numpy(p).cuda()
    d_out = torch.from_numpy(init.copy()).cuda()
    d_ret = torch.full((nsf,), 12345, dtype=torch.int32, device="cuda")
    V.rs_batch_dev(d_p, d_out, d_ret, rsdims, nsf)
    torch.cuda.synchronize()
    assert np.array_equal(d_ret.cpu().numpy(), ret_ref)
    assert np.array_equal(d_out.cpu().numpy(), out_ref)


@pytest.mark.parametrize("rsdims,weights", [
    (24, {0: 90, 1: 5, 2: 2, 3: 1, 4: 1, 6: 1}),            # a few high-degree columns per wave, some uncorrectable
    (24, {0: 40, 1: 10, 2: 10, 3: 15, 5: 15, 6: 8, 9: 2}),  # about a third of the lanes, failures end superframes early
    (24, {3: 30, 4: 30, 5: 38, 7: 2}),                      # every lane
    (4, {0: 70, 2: 10, 3: 10, 5: 5, 6: 5}),                 # 16 superframes per wave
    (64, {0: 80, 3: 12, 5: 5, 8: 3}),                       # a superframe = one wave
    (256, {0: 85, 1: 5, 4: 8, 6: 2}),                       # a superframe = the whole workgroup
    (3, {0: 60, 1: 20, 3: 15, 6: 5}),                       # superframes straddle waves (85 per pass)
    (1, {0: 70, 3: 20, 6: 10}),                             # 128 superframes per pass
])
def test_rs_error_mixes(V, O, torch_cuda, rsdims, weights):
    """The locator-root search has three forms (closed form, a wavefront per column with the rest of a failed
    superframe dropped, a column per lane); these mixes reach all of them and the hand-over between them."""
    torch = torch_cuda
    rng = np.random.default_rng(1000 + rsdims + len(weights))
    nsf = max(24, 3072 // rsdims)
    kinds = np.array(list(weights.keys())); pr = np.array(list(weights.values()), float); pr /= pr.sum()
    p = np.empty((nsf, 120, rsdims), np.uint8)
    for s_ in range(nsf):
        for j in range(rsdims):
            cw = O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8))
            ne = int(rng.choice(kinds, p=pr))
            pos = rng.choice(120, ne, replace=False)
            cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
            p[s_, :, j] = cw
    p = p.reshape(nsf, 120 * rsdims)
    init = np.full((nsf, 110 * rsdims), 0xA5, np.uint8)
    ret_ref, out_ref = O.rs_check_batch(p, rsdims, out_init=init)
    d_p = torch.from_numpy(p).cuda()
    d_out = torch.from_numpy(init.copy()).cuda()
    d_ret = torch.full((nsf,), 12345, dtype=torch.int32, device="cuda")
    V.rs_batch_dev(d_p, d_out, d_ret, rsdims, nsf)
    torch.cuda.synchronize()
    assert np.array_equal(d_ret.cpu().numpy(), ret_ref)
    assert np.array_equal(d_out.cpu().numpy(), out_ref)


@pytest.mark.parametrize("rsdims,in_off,out_off", [(8, 0, 0), (8, 8, 16), (8, 4, 4), (8, 1, 0), (8, 0, 1), (6, 0, 0),
                                                    (6, 2, 2), (7, 0, 0), (7, 3, 5), (2, 0, 2), (24, 8, 4), (24, 16, 8)])
def test_rs_pointer_alignment(V, O, torch_cuda, rsdims, in_off, out_off):
    """The kernel picks its copy widths from the block sizes and the alignment of the device pointers (16-byte, dword
    and byte paths in both directions): same results whatever they are, and not a byte outside the output block."""
    torch = torch_cuda
    nsf = 37
    p = _rs_superframes(O, nsf, rsdims, seed=77 + rsdims + in_off)
    init = np.full((nsf, 110 * rsdims), 0xC3, np.uint8)
    ret_ref, out_ref = O.rs_check_batch(p, rsdims, out_init=init)
    pad = 64
    d_in = torch.zeros(p.size + 2 * pad, dtype=torch.uint8, device="cuda")
    d_in[in_off:in_off + p.size] = torch.from_numpy(p.reshape(-1)).cuda()
    d_o = torch.full((init.size + 2 * pad,), 0xC3, dtype=torch.uint8, device="cuda")
    d_ret = torch.full((nsf,), 12345, dtype=torch.int32, device="cuda")
    V.rs_batch_dev(d_in[in_off:], d_o[pad + out_off:], d_ret, rsdims, nsf)
    torch.cuda.synchronize()
    got = d_o.cpu().numpy()
    assert np.array_equal(d_ret.cpu().numpy(), ret_ref)
    assert np.array_equal(got[pad + out_off:pad + out_off + init.size].reshape(nsf, -1), out_ref)
    assert (got[:pad + out_off] == 0xC3).all() and (got[pad + out_off + init.size:] == 0xC3).all()


def test_rs_wide_superframe(V, O, torch_cuda):
    """more columns than a workgroup has lanes: chunked walk keeps the early-exit rule"""
    rsdims, nsf = 300, 3
    p = _rs_superframes(O, nsf, rsdims, seed=9, max_err=5)
    p = p.reshape(nsf, 120, rsdims)
    p[1, :8, 270] ^= 0x3C  # 8 errors in column 270 of superframe 1 -> -1 in the second chunk
    p = p.reshape(nsf, -1)
    init = np.full((nsf, 110 * rsdims), 0x5A, np.uint8)
    ret_ref, out_ref = O.rs_check_batch(p, rsdims, out_init=init)
    ret, out = V.rs_batch_host(p, rsdims, out_init=init)
    assert ret_ref[1] == -1
    assert np.array_equal(ret, ret_ref) and np.array_equal(out, out_ref)


def test_rscheck_export(V, O, torch_cuda):
    rsdims = 12
    p = np.zeros(120 * rsdims, np.uint8)  # SURVEY 8c KAT: zeros with 3 flipped bytes -> 3, zeros restored
    p[3], p[500], p[1300] = 0x55, 0x01, 0xFF
    rc, out = V.RScheckSuperframe(p, 0, rsdims)
    assert rc == 3 and not out.any()
    q = np.zeros(120 * rsdims, np.uint8)
    q[[5 + rsdims * k for k in (1, 9, 20, 33, 47, 90)]] = [7, 99, 3, 200, 5, 66]  # 6 errors in column 5 -> -1
    sentinel = np.full(110 * rsdims, 0x77, np.uint8)
    rc, out = V.RScheckSuperframe(q, 0, rsdims, sentinel.copy())
    rc_ref, out_ref = O.rs_check_superframe(q, rsdims, sentinel.copy())
    assert rc == rc_ref == -1 and np.array_equal(out, out_ref)
    assert V.lib().RSCheckSuperframe(None, 0, 0, None) == 0
    assert V.lib().RScheckSuperframe(None, 0, 10, None) == -1


def test_dabplus_superframe_pipeline(V, O, torch_cuda):
    """BASELINE config 5 in small: RS-coded payload -> mother code -> noise -> deconvolve x5 ->
    RScheckSuperframe, GPU pipeline against the oracle's two stages."""
    torch = torch_cuda
    rsdims, nsf = 12, 24
    fb = 192 * rsdims
    rng = np.random.default_rng(12)
    syms = np.empty((nsf * 5, O.sym_len(fb)), np.uint8)
    state = 1234567
    for s in range(nsf):
        block = np.empty((120, rsdims), np.uint8)
        for j in range(rsdims):
            cw = O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8))
            ne = int(rng.choice([0, 0, 1, 3, 5] if s % 2 == 0 else [0, 0, 1, 3, 5, 6, 7]))  # post-Viterbi symbol errors
            pos = rng.choice(120, ne, replace=False)
            cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
            block[:, j] = cw
        bits = np.unpackbits(block.reshape(-1)).reshape(5, fb)
        for k in range(5):
            hard = O.encode(bits[k]).astype(np.float64)
            noise = rng.normal(0.0, 1.0, hard.size)
            v = 127.5 + 32.0 * ((hard * 2 - 1) * 1.6 + noise)  # mild noise: a few residual bit errors at most
            syms[s * 5 + k] = np.clip(v.astype(np.int64), 0, 255).astype(np.uint8)
    dec_ref = O.decode_batch(fb, syms, nthreads=8).reshape(nsf, 120 * rsdims)
    init = np.full((nsf, 110 * rsdims), 0x3C, np.uint8)
    ret_ref, out_ref = O.rs_check_batch(dec_ref, rsdims, out_init=init)
    assert (ret_ref == -1).any() and (ret_ref >= 0).any()
    d_sym = torch.from_numpy(syms).cuda()
    d_work = torch.zeros((nsf, 120 * rsdims), dtype=torch.uint8, device="cuda")
    d_out = torch.from_numpy(init.copy()).cuda()
    d_ret = torch.zeros(nsf, dtype=torch.int32, device="cuda")
    V.dabplus_superframes_dev(d_sym, d_work, d_out, d_ret, rsdims, nsf)
    torch.cuda.synchronize()
    assert np.array_equal(d_work.cpu().numpy(), dec_ref)
    assert np.array_equal(d_ret.cpu().numpy(), ret_ref)
    assert np.array_equal(d_out.cpu().numpy(), out_ref)


def test_ber_harness_matches_oracle(V, O, torch_cuda):
    """viterbi-benchmark.cpp:293-329 analogue: at Eb/N0 = 3 dB the decoder is nearly error free, and
    the GPU's error count equals the oracle's on the same frames (it is the same bits)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vit_ber", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "ber.py"))
    ber = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ber)
    r = ber.run(3.0, 400, 3072, seed=5)
    assert r["ber"] < 2e-3
    import bench
    sym, bits = bench.make_frames(400, 3072, seed=5, device=torch_cuda.device("cuda:0"), return_bits=True)
    ref = np.unpackbits(O.decode_batch(3072, sym.cpu().numpy(), nthreads=8), axis=1)
    assert int((ref != bits.cpu().numpy()).sum()) == r["bit_errors"]
    hard = ber.run(20.0, 50, 768, seed=1)  # essentially noise free
    assert hard["bit_errors"] == 0


# ---- BASELINE sizes through size-independent properties ----------------------------

def test_full_size_fic_batch_properties(V, O, torch_cuda):
    """BASELINE config 2 (65536 FIC frames): the batch is 256 distinct frames tiled 256x, so
    every tile must equal the oracle's decode of the 256 (checksum of checksums)."""
    torch = torch_cuda
    framebits, base_n, reps = 768, 256, 256
    base = _mixed_input(O, base_n, framebits, seed=2024)
    want = O.decode_batch(framebits, base, nthreads=8)
    d_base = torch.from_numpy(base).cuda()
    d_sym = d_base.repeat(reps, 1).contiguous()
    n = base_n * reps
    d_out = torch.zeros((n, framebits // 8), dtype=torch.uint8, device="cuda")
    V.decode_batch_dev(d_sym, d_out, framebits, n)
    torch.cuda.synchronize()
    got = d_out.view(reps, base_n, framebits // 8)
    d_want = torch.from_numpy(want).cuda()
    assert bool((got == d_want.unsqueeze(0)).all())


def test_full_size_fic_batch_ge_mode(V, O, torch_cuda):
    """BASELINE config 2 (65536 FIC frames) with the MASM decoders' `>= 150` comparator (vit_set_renorm_ge(1)): the
    batch is tiled from 256 distinct frames - soft-decision ones and the two hard-decision families on which the two
    comparators give different outputs (asserted) - and every tile must equal the ge oracle's decode."""
    torch = torch_cuda
    framebits, base_n, reps = 768, 256, 256
    base = np.concatenate([_mixed_input(O, base_n // 2, framebits, seed=77), _hard_families(O, framebits, base_n // 4, seed=78)])
    assert base.shape[0] == base_n
    want_ge = O.decode_batch(framebits, base, nthreads=8, ge=True)
    want_gt = O.decode_batch(framebits, base, nthreads=8)
    assert (want_ge != want_gt).any(axis=1).sum() >= 2  # the mode matters on this batch
    d_sym = torch.from_numpy(base).cuda().repeat(reps, 1).contiguous()
    n = base_n * reps
    old = V.set_renorm_ge(1)
    try:
        d_out = torch.zeros((n, framebits // 8), dtype=torch.uint8, device="cuda")
        V.decode_batch_dev(d_sym, d_out, framebits, n)
        torch.cuda.synchronize()
    finally:
        V.set_renorm_ge(old)
    assert bool((d_out.view(reps, base_n, framebits // 8) == torch.from_numpy(want_ge).cuda().unsqueeze(0)).all())
