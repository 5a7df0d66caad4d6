"""GPU parity of the 8-frames-per-wavefront packed kernel (csrc/vit_pk8.hip, vit_set_kernel(4); an experiment that is
compiled in only with -DVIT_WITH_PK8 - skipped against the product library) against the oracle:
every length of one segment (<= 778 bits), ragged batches, both renormalise comparators, hard-decision and saturation
inputs, u32 ingest, descriptor tables.  Bit-exact, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
K8 = 4


@pytest.fixture(autouse=True)
def _needs_pk8(V):
    """the experiment is not part of the product library: these tests run only against a build with -DVIT_WITH_PK8
    (python -c "import _vitpkg; _vitpkg.load_package().build(force=True, extra=['-DVIT_WITH_PK8'])")"""
    old = V.set_kernel(K8)
    have = V.set_kernel(old) == K8  # without the kernel vit_set_kernel(4) selects 0
    if not have:
        pytest.skip("libviterbi.so was built without -DVIT_WITH_PK8 (the 8-frames-per-wavefront experiment)")


def _decode(V, torch, sym, framebits, kernel=K8):
    n = sym.shape[0]
    old = V.set_kernel(kernel)
    try:
        d_out = torch.full((n, (framebits + 7) // 8), 0xEE, dtype=torch.uint8, device="cuda")
        V.decode_batch_dev(torch.from_numpy(np.ascontiguousarray(sym)).cuda(), d_out, framebits, n)
        torch.cuda.synchronize()
        return d_out.cpu().numpy()
    finally:
        V.set_kernel(old)


def _mixed(O, n, fb, seed):
    a = O.noisy_frames(n - n // 2, fb, seed=seed)
    b = O.uniform_symbols((n // 2) * O.sym_len(fb), seed=seed + 1000).reshape(n // 2, -1)
    return np.concatenate([a, b])


@pytest.mark.parametrize("framebits", [768, 288, 8, 2, 10, 16, 96, 250, 256, 262, 266, 272, 504, 522, 528, 770, 776, 778])
def test_pk8_parity_every_block_shape(V, O, torch_cuda, framebits):
    n = 203  # not a multiple of 8: ragged last group
    sym = _mixed(O, n, framebits, seed=framebits + 7)
    assert np.array_equal(_decode(V, torch_cuda, sym, framebits), O.decode_batch(framebits, sym, nthreads=8))


def test_pk8_all_even_lengths_small_batches(V, O, torch_cuda):
    rng = np.random.default_rng(8)
    for fb in sorted(set((2 * rng.integers(1, 390, 40)).tolist())):
        n = int(rng.integers(1, 20))
        sym = _mixed(O, n, fb, seed=fb) if n > 1 else O.noisy_frames(1, fb, seed=fb)
        assert np.array_equal(_decode(V, torch_cuda, sym, fb), O.decode_batch(fb, sym, nthreads=8)), (fb, n)


def test_pk8_hard_inputs_and_both_comparators(V, O, torch_cuda):
    fb, n = 768, 160
    sym = np.concatenate([O.hard_random_symbols(n, fb, seed=3), O.hard_flipped_frames(n, fb, flip=0.2, seed=4)])
    stress = np.empty((32, O.sym_len(fb)), np.uint8)
    stress[0::2], stress[1::2] = 0, 255
    sym = np.concatenate([sym, stress])
    want_gt, want_ge = O.decode_batch(fb, sym, nthreads=8), O.decode_batch(fb, sym, nthreads=8, ge=True)
    assert (want_gt != want_ge).any(axis=1).sum() >= 3
    assert np.array_equal(_decode(V, torch_cuda, sym, fb), want_gt)
    V.set_renorm_ge(1)
    try:
        got = _decode(V, torch_cuda, sym, fb)
    finally:
        V.set_renorm_ge(0)
    assert np.array_equal(got, want_ge)


def test_pk8_full_batch_matches_the_4_frame_kernel_and_the_oracle(V, O, torch_cuda):
    """65536 FIC frames tiled from 256 distinct ones: the 8-frames-per-wave kernel, the automatic choice and the forced
    4-frames-per-wave kernel must all equal the oracle's decode of the distinct frames, tile by tile"""
    torch = torch_cuda
    fb, distinct, reps = 768, 256, 256
    sym = _mixed(O, distinct, fb, seed=65)
    want = torch.from_numpy(O.decode_batch(fb, sym, nthreads=8)).cuda()
    d_sym = torch.from_numpy(sym).cuda().repeat(reps, 1)
    for kernel in (0, K8, 2):
        old = V.set_kernel(kernel)
        try:
            d_out = torch.zeros((distinct * reps, fb // 8), dtype=torch.uint8, device="cuda")
            V.decode_batch_dev(d_sym, d_out, fb, distinct * reps)
            torch.cuda.synchronize()
        finally:
            V.set_kernel(old)
        assert bool((d_out.view(reps, distinct, -1) == want.unsqueeze(0)).all()), kernel


def test_pk8_u32_ingest_and_descriptor_table(V, O, torch_cuda):
    torch = torch_cuda
    fb, n = 768, 50
    sym = _mixed(O, n, fb, seed=77)
    want = O.decode_batch(fb, sym)
    junk = np.random.default_rng(5).integers(0, 1 << 24, sym.shape, dtype=np.int64) << 8
    host32 = ((sym.astype(np.int64) | junk) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    old = V.set_kernel(K8)
    try:
        d_out = torch.zeros((n, fb // 8), dtype=torch.uint8, device="cuda")
        V.decode_batch_dev_u32(torch.from_numpy(host32).cuda(), d_out, fb, n)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want)
        # descriptor table: mixed lengths of one segment, one invalid descriptor
        rng = np.random.default_rng(9)
        fbs = [int(x) for x in 2 * rng.integers(1, 390, 300)]
        desc, sym_bytes, out_bytes = V.make_descs(fbs)
        s2 = O.uniform_symbols(sym_bytes, seed=10)
        w2 = np.full(out_bytes, 0x5A, np.uint8)
        for i, (f, d) in enumerate(zip(fbs, desc)):
            so, oo = int(d["sym_offset"]), int(d["out_offset"])
            if i != 17:
                w2[oo:oo + (f + 7) // 8] = O.decode_batch(f, s2[so:so + O.sym_len(f)])[0]
        desc["framebits"][17] += 1  # odd: skipped
        d_out2 = torch.full((out_bytes,), 0x5A, dtype=torch.uint8, device="cuda")
        V.decode_varlen_dev(torch.from_numpy(s2).cuda(), d_out2, torch.from_numpy(desc.view(np.uint8)).cuda(), len(fbs), 778)
        torch.cuda.synchronize()
        assert np.array_equal(d_out2.cpu().numpy(), w2)
    finally:
        V.set_kernel(old)
