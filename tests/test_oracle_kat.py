"""CPU-only: pins the oracle (test infrastructure) against every known-answer vector that
exists for this path, and the oracle's two implementations against each other.

What pins it: SURVEY.md section 8c's probe KATs, i.e. outputs of the compiled reference taken
during the survey (the reference cannot be built under this round's rules: it needs
stand-ins for <windows.h>/<psapi.h> and MASM constants).  The 16-byte decoder prefix, the
GF table samples and the two RS behaviours reproduce.  The survey's three FNV-1a-64
digests were wrong; the full-length digests the round-1 judge obtained from a stand-in build of the
reference reproduce (test_full_length_digests_from_the_round1_judge_probe, informational).
"""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_kat_decoder_prefix(O):
    # sym[i] = (xorshift64(13,7,17) >> 11) & 255, seed 88172645463325252; the first 16
    # output bytes are the same for framebits 288 / 768 / 6912
    want = bytes.fromhex("fa86dfa7873fb335820f0977ead397f7")
    for fb in (288, 768, 6912):
        sym = O.uniform_symbols(O.sym_len(fb))
        out = O.deconvolve_u32(fb, sym.astype(np.uint32))
        assert out[:16].tobytes() == want
        assert O.decode_batch(fb, sym)[0][:16].tobytes() == want
        if O.has_avx2():
            assert O.decode_batch(fb, sym, avx2=True)[0][:16].tobytes() == want


def test_full_length_digests_from_the_round1_judge_probe(O):
    """INFORMATIONAL, not a pin.  SURVEY 8c's three FNV-1a-64 digests were wrong (they never reproduced, although
    the 16-byte prefix from the same run did).  The round-1 judge compiled the reference's deconvolve.cpp in /tmp
    with stand-ins for <windows.h>/<psapi.h> and the MASM constants - a stand-in build, which under this project's
    rules pins nothing - and reported the digests of ITS outputs on SURVEY 8c's input:
    those are the values below, and this oracle reproduces all three over the full output length."""
    probe = {288: 0xA2C99CA2ACF1194F, 768: 0x5CD1C7D0DEC24659, 6912: 0x87E1C4A6BA2838F5}
    for fb, dig in probe.items():
        sym = O.uniform_symbols(O.sym_len(fb)).astype(np.uint32)
        assert O.fnv1a64(O.deconvolve_u32(fb, sym)) == dig
        if O.has_avx2():
            assert O.fnv1a64(O.decode_batch(fb, sym.astype(np.uint8), avx2=True)[0]) == dig


def test_kat_gf_tables(O):
    ato, iof = O.rs_tables()
    assert ato[:10].tolist() == [1, 2, 4, 8, 16, 32, 64, 128, 29, 58]
    assert iof[1:6].tolist() == [0, 1, 25, 2, 50]
    assert iof[0] == 255 and ato.size == 768 and (ato[255:510] == ato[:255]).all()
    assert (ato[iof[1:].astype(int)] == np.arange(1, 256)).all()


def test_kat_rs_behaviour(O):
    # all-zero 120x12 block with 3 flipped bytes -> 3 and zeros restored
    p = np.zeros(120 * 12, np.uint8)
    p[3], p[500], p[1300] = 0x55, 0x01, 0xFF
    rc, out = O.rs_check_superframe(p, 12)
    assert rc == 3 and not out.any()
    # 6 errors in one column -> -1, and nothing at/after that column is written
    q = np.zeros(120 * 12, np.uint8)
    q[[5 + 12 * k for k in (1, 9, 20, 33, 47, 90)]] = [7, 99, 3, 200, 5, 66]
    sentinel = np.full(110 * 12, 0x77, np.uint8)
    rc, out = O.rs_check_superframe(q, 12, sentinel.copy())
    assert rc == -1
    out = out.reshape(110, 12)
    assert (out[:, 5:] == 0x77).all() and (out[:, :5] == 0).all()


def test_rs_corrects_up_to_five(O):
    rng = np.random.default_rng(0)
    for ne in range(0, 6):
        for _ in range(20):
            msg = rng.integers(0, 256, 110, dtype=np.uint8)
            cw = O.rs_encode(msg)
            rc0, _ = O.rs_decode_word(cw)
            assert rc0 == 0
            bad = cw.copy()
            pos = rng.choice(120, ne, replace=False)
            bad[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
            rc, fixed = O.rs_decode_word(bad)
            assert rc == ne and np.array_equal(fixed[:110], msg)


def test_scalar_equals_avx2_port(O):
    if not O.has_avx2():
        pytest.skip("host without AVX2")
    for fb in (8, 96, 288, 768, 1536, 3072, 6912, 9216):
        n = 24 if fb <= 3072 else 6
        sym = np.concatenate([O.noisy_frames(n // 2, fb, seed=fb),
                              O.uniform_symbols((n // 2) * O.sym_len(fb), seed=fb + 1).reshape(n // 2, -1)])
        assert np.array_equal(O.decode_batch(fb, sym), O.decode_batch(fb, sym, avx2=True))


def test_u32_entry_uses_low_byte_only(O):
    fb = 768
    sym = O.uniform_symbols(O.sym_len(fb), seed=2)
    a = O.deconvolve_u32(fb, sym.astype(np.uint32))
    b = O.deconvolve_u32(fb, sym.astype(np.uint32) | np.uint32(0xDEAD0000))  # deconvolve.cpp:158-165
    assert np.array_equal(a, b) and np.array_equal(a, O.decode_batch(fb, sym)[0])


def test_renorm_comparator_ge_vs_gt(O):
    """Appendix A.6: the MASM twins renormalise on >=150 (decon_avx2.asm:97,114), the C path on >150
    (deconvolve.cpp:408).  On soft-decision input (reference-style noise, uniform bytes) the two modes agree --
    the survey's 36000-frame observation -- but on HARD-decision input they do not: a uniform -63 is invisible
    only until a metric hits the 0 or the 255 clamp (round-2 judge: 14-54 % of such frames differ)."""
    fb = 768
    soft = np.concatenate([O.noisy_frames(40, fb, seed=3), O.uniform_symbols(40 * O.sym_len(fb), seed=4).reshape(40, -1)])
    assert np.array_equal(O.decode_batch(fb, soft), O.decode_batch(fb, soft, ge=True))
    for s in soft[:4]:  # the u32 entry point takes the same switch
        assert np.array_equal(O.deconvolve_u32(fb, s.astype(np.uint32), ge=True), O.decode_batch(fb, s, ge=True)[0])
    differing = {}
    for fb, n in ((768, 100), (3072, 60)):
        for name, sym in (("random 0/255", O.hard_random_symbols(n, fb, seed=5)),
                          ("encoded, 20 % flips", O.hard_flipped_frames(n, fb, flip=0.2, seed=5))):
            differing[(fb, name)] = int((O.decode_batch(fb, sym, nthreads=4) != O.decode_batch(fb, sym, nthreads=4, ge=True))
                                        .any(axis=1).sum())
    assert all(v > 0 for v in differing.values()), differing
    assert differing[(3072, "encoded, 20 % flips")] >= 20  # more than a third of these frames


def test_noise_free_roundtrip_and_ber(O):
    rng = np.random.default_rng(1)
    for fb in (8, 768, 3072):
        bits = rng.integers(0, 2, fb, dtype=np.uint8)
        sym = (O.encode(bits) * 255).astype(np.uint8)
        assert np.array_equal(np.unpackbits(O.decode_batch(fb, sym)[0]), bits)
    # reference-style BER check (viterbi-benchmark.cpp:296-329): Eb/N0 = 3 dB must decode nearly clean
    sym, bits = O.noisy_frames(50, 3072, seed=9, return_bits=True)
    out = np.unpackbits(O.decode_batch(3072, sym, nthreads=4), axis=1)
    ber = (out != bits).mean()
    assert ber < 2e-3


# --- independent cross-check of the natural-order derivation (SURVEY Appendix A.3 from A.4) ---
# The 96 mask bytes below are DATA: the lane-permuted 256-bit constants exactly as const.asm:27-63
# lays them out.  The step function works in that permuted order, like the reference's AVX2 path.
_M256_0347 = bytes.fromhex("0000ffffffff0000ffff00000000ffff" * 2)
_M256_15 = bytes.fromhex("00ffff00ff0000ff00ffff00ff0000ff" * 2)
_M256_26 = bytes.fromhex("00ff00ff00ff00ff00ff00ff00ff00ff" + "ff00ff00ff00ff00ff00ff00ff00ff00")


def _decode_permuted_order(fb, sym):
    k0 = np.frombuffer(_M256_0347, np.uint8).astype(np.int32)
    k1 = np.frombuffer(_M256_15, np.uint8).astype(np.int32)
    k2 = np.frombuffer(_M256_26, np.uint8).astype(np.int32)
    avg = lambda a, b: (a + b + 1) >> 1
    perm = np.r_[0:8, 16:24, 8:16, 24:32]  # qword order [0,2,1,3]: position -> state (vector A)
    A = np.full(32, 63, np.int32)
    A[0] = 0
    B = np.full(32, 63, np.int32)
    T = ((fb + 6) // 2) * 2
    dec = np.zeros(T, np.uint64)
    for t in range(T):
        s = sym[4 * t:4 * t + 4].astype(np.int32)
        met = avg(avg(s[0] ^ k0, s[1] ^ k1), avg(s[2] ^ k2, s[3] ^ k0)) >> 2
        mm = 63 - met
        m0, m1 = np.minimum(A + met, 255), np.minimum(B + mm, 255)
        m2, m3 = np.minimum(A + mm, 255), np.minimum(B + met, 255)
        d0, d1 = m1 <= m0, m3 <= m2
        sv0, sv1 = np.where(d0, m1, m0), np.where(d1, m3, m2)
        new = np.zeros(64, np.int32)
        word = 0
        for pos in range(32):
            i = int(perm[pos])  # butterfly held at this position
            new[2 * i], new[2 * i + 1] = sv0[pos], sv1[pos]
            word |= (int(d0[pos]) << (2 * i)) | (int(d1[pos]) << (2 * i + 1))
        dec[t] = word
        A, B = new[perm], new[perm + 32]
        if (t & 1) and A[0] > 150:
            A, B = np.maximum(A - 63, 0), np.maximum(B - 63, 0)
    out = np.zeros((fb + 7) // 8, np.uint8)
    E = 0
    for n in range(fb - 1, -1, -1):
        k = (int(dec[n + 6]) >> (E >> 2)) & 1
        E = ((E >> 1) | (k << 7)) & 0xFF
        out[n >> 3] = E
    return out


def test_permuted_order_restatement_agrees(O):
    for fb, seed in ((96, 1), (288, 2), (768, 3)):
        for sym in (O.noisy_frames(1, fb, seed=seed)[0], O.uniform_symbols(O.sym_len(fb), seed=seed)):
            assert np.array_equal(_decode_permuted_order(fb, sym), O.decode_batch(fb, sym)[0])


# --- committed regression fixtures (generated by this oracle, tests/golden/make_golden.py) ---

def _golden_sym(case):
    import base64
    import zlib
    return np.frombuffer(zlib.decompress(base64.b64decode(case["sym_zb64"])), np.uint8)


def test_golden_fixtures(O):
    with open(os.path.join(GOLD, "golden.json")) as f:
        g = json.load(f)
    assert len(g["decode"]) >= 17 and len(g["rs"]) >= 10
    separating = 0
    for case in g["decode"]:
        fb = case["framebits"]
        sym = _golden_sym(case)
        assert sym.size == O.sym_len(fb) and O.fnv1a64(sym) == int(case["sym_fnv1a64"], 16)
        if case["kind"] == "uniform":  # the stored bytes are the SURVEY KAT generator's stream
            assert np.array_equal(sym, O.uniform_symbols(O.sym_len(fb), seed=case["seed"]))
        assert O.decode_batch(fb, sym)[0].tobytes().hex() == case["out_hex"]
        assert O.decode_batch(fb, sym, ge=True)[0].tobytes().hex() == case["out_ge_hex"]  # the MASM comparator
        if fb > 0 and fb % 8 == 0 and O.has_avx2():
            assert O.decode_batch(fb, sym, avx2=True)[0].tobytes().hex() == case["out_hex"]
        separating += case["out_hex"] != case["out_ge_hex"]
    assert separating >= 5
    assert {6912, 9216, 770, 2} <= {c["framebits"] for c in g["decode"]}
    for case in g["rs"]:
        p = np.frombuffer(bytes.fromhex(case["p_hex"]), np.uint8)
        rc, out = O.rs_check_superframe(p, case["rsdims"], np.full(110 * case["rsdims"], 0xA5, np.uint8))
        assert rc == case["ret"] and out.tobytes().hex() == case["out_hex"]
    assert {24, 16} <= {c["rsdims"] for c in g["rs"]} and any(c["pad_cols"] for c in g["rs"])


def test_packed_layout_emulation(O):
    """CPU-only check of the packed kernel's index math (tests/tools/emulate_pk.py): rotating lane<->state
    map, table triples, decision-history layout and the traceback position formula, emulated lane by
    lane in numpy, must reproduce the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "emulate_pk", os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "emulate_pk.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    for fb in (40, 136):
        sym = np.concatenate([O.noisy_frames(2, fb, seed=fb), O.uniform_symbols(2 * O.sym_len(fb), seed=fb + 1).reshape(2, -1)])
        assert np.array_equal(emu.emulate(sym, fb), O.decode_batch(fb, sym))


def test_rs_syndromes_from_generator_remainder():
    """CPU check of the identity csrc/rs_kernels.hip relies on: the ten syndromes of the reference's
    Horner loop (rschecksf.cpp:210-219) equal the remainder of the codeword modulo
    g(x) = prod_{i=0..9}(x + alpha^i) evaluated at alpha^i; a valid codeword has remainder 0."""
    alpha, iof, sr = [0] * 255, [255] * 256, 1
    for i in range(255):  # dllmain.cpp:124-146
        iof[sr], alpha[i] = i, sr
        sr <<= 1
        if sr & 256:
            sr ^= 285
        sr &= 255

    def mul(a, b):
        return 0 if a == 0 or b == 0 else alpha[(iof[a] + iof[b]) % 255]

    g = [1]
    for i in range(10):
        ng = [0] * (len(g) + 1)
        for j, c in enumerate(g):
            ng[j + 1] ^= c
            ng[j] ^= mul(c, alpha[i])
        g = ng
    assert g[10] == 1 and g[:10] == [193, 157, 113, 95, 94, 199, 111, 159, 194, 216]
    rng = np.random.default_rng(9)
    import _vitpkg
    O = _vitpkg.load_oracle()
    for trial in range(40):
        d = rng.integers(0, 256, 120).tolist() if trial % 2 else O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8)).tolist()
        S = [d[0]] * 10
        for k in range(1, 120):
            S = [mul(S[i], alpha[i]) ^ d[k] for i in range(10)]
        r = [0] * 10
        for k in range(120):  # r <- r*x + d_k mod g
            f = r[9]
            r = [d[k] ^ mul(f, g[0])] + [r[j - 1] ^ mul(f, g[j]) for j in range(1, 10)]
        S2 = []
        for i in range(10):
            v = r[9]
            for j in range(8, -1, -1):
                v = mul(v, alpha[i]) ^ r[j]
            S2.append(v)
        assert S == S2
        assert (trial % 2 == 1) or (r == [0] * 10 and S == [0] * 10)


def test_renormalise_comparators_on_a_hand_derived_trajectory(O):
    """Pins the oracle's TWO renormalise comparators by a trajectory derived by hand from the source lines, not by the
    oracle's own code (round-3 advisor).  Input: every soft symbol = 100 (a weak "0" on all four outputs).

    * Branch metric (deconvolve.cpp:335-351): x_j = 100 or 155 (= 100 ^ 0xFF); the all-zero branch has x = (100,100,100,100),
      metric = avg(avg(100,100),avg(100,100)) >> 2 = 25; any other class holds at least one 155, so metric >= 25, and
      since metric <= 155 >> 2 = 38 also 63 - metric >= 25: EVERY branch costs at least 25.
    * So after t+1 steps every state's metric is >= 25 (t+1) minus the uniform renormalisations (initial metrics are
      0 and 63, const.asm:19-25), and the all-zero path (state 0 -> state 0, butterfly 0: new[0] = min(old[0] + 25,
      old[32] + 38)) attains it: m0(t) = 25 (t+1) - 63 r(t), r = renormalisations so far.  No clamp interferes: m0 + 25
      <= 225 < 255, and a renormalisation happens at m0 >= 150 > 63 with every other metric >= m0.
    * Renormalise256 runs after the odd steps t = 1, 3, 5, ... (deconvolve.cpp:398-412).  m0 = 50, 100, then EXACTLY 150
      at t = 5: `> 150` (C decoders, deconvolve.cpp:408) does not subtract, `>= 150` (MASM decoders,
      decon_avx2.asm:97,114 `cmp sil,150 ; jb`) does.  From t = 9 on the two trajectories coincide again."""
    fb = 18  # 24 trellis steps
    sym = np.full(O.sym_len(fb), 100, np.uint8)
    want_gt = [25, 50, 75, 100, 125, 150, 175, 200 - 63, 162, 187 - 63, 149, 174 - 63, 136, 161 - 63, 123, 148, 173, 198 - 63]
    want_ge = [25, 50, 75, 100, 125, 150 - 63, 112, 137, 162, 187 - 63, 149, 174 - 63, 136, 161 - 63, 123, 148, 173, 198 - 63]
    assert O.trace_state0(fb, sym, ge=False)[:18].tolist() == want_gt
    assert O.trace_state0(fb, sym, ge=True)[:18].tolist() == want_ge
    assert want_gt[5:9] != want_ge[5:9] and want_gt[9:] == want_ge[9:]
    # the decoded bits are the all-zero message in both modes (the uniform shift is invisible without a clamp)
    assert not O.decode_batch(fb, sym[None, :]).any() and not O.decode_batch(fb, sym[None, :], ge=True).any()
