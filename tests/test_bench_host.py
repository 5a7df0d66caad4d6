"""CPU checks of bench.py's host logic: the synthetic-input generator follows the reference recipe
(viterbi-benchmark.cpp:293-311) - its noiseless symbols are the oracle encoder's code bits, the payload
hook reproduces given bits - and the sharding arithmetic used for N > 1."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_make_frames_matches_oracle_encoder(O):
    fb, n = 96, 6
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2, (n, fb)).astype(np.int32)
    # Eb/N0 = 60 dB: the noise term vanishes, samples saturate at 0 / 255 exactly where the code bit is 0 / 1
    sym = bench.make_frames(n, fb, seed=1, device=torch.device("cpu"), ebn0_db=60.0,
                            payload_bits=torch.from_numpy(bits)).numpy()
    assert sym.shape == (n, 4 * (fb + bench.TAIL)) and sym.dtype == np.uint8
    for i in range(n):
        hard = O.encode(bits[i].astype(np.uint8))
        assert np.array_equal(sym[i] > 127, hard.astype(bool))
        # and the oracle decodes it back (tail-terminated frame)
        assert np.array_equal(np.unpackbits(O.decode_batch(fb, sym[i:i + 1])[0]), bits[i].astype(np.uint8))


def test_make_frames_is_seeded_and_noisy():
    a = bench.make_frames(4, 768, seed=5, device=torch.device("cpu"))
    b = bench.make_frames(4, 768, seed=5, device=torch.device("cpu"))
    c = bench.make_frames(4, 768, seed=6, device=torch.device("cpu"))
    assert torch.equal(a, b) and not torch.equal(a, c)
    v = a.to(torch.float32)
    assert 20 < float(v.std()) < 120 and 0 < int((a == 0).sum()) and 0 < int((a == 255).sum())  # clipped AWGN around 127.5


def test_bench_constants():
    assert bench.FRAMEBITS == 768 and bench.TAIL == 6 and bench.POLYS == (109, 79, 83, 109)
    assert 4 * (bench.FRAMEBITS + bench.TAIL) + bench.FRAMEBITS // 8 == 3192  # algorithmic bytes per FIC frame
