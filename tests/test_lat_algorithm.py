"""CPU check of the latency kernel's ALGORITHM (csrc/vit_lat.hip): tests/tools/emulate_lat.py replays its rotating
lane <-> state map, partner fetch, class function, decision rule (tie -> 1 on both sides of a butterfly) and blocked
speculative traceback with numpy vectors standing in for the 64 lanes; the result must equal the oracle's."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    spec = importlib.util.spec_from_file_location("emulate_lat", os.path.join(ROOT, "tests", "tools", "emulate_lat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("fb,kind", [(96, "noisy"), (96, "uniform"), (10, "uniform"), (288, "noisy"), (770, "uniform")])
def test_emulated_latency_kernel_matches_oracle(O, emu, fb, kind):
    sym = O.noisy_frames(1, fb, seed=fb)[0] if kind == "noisy" else O.uniform_symbols(O.sym_len(fb), seed=fb)
    assert np.array_equal(emu.decode(sym, fb), O.decode_batch(fb, sym)[0])


def test_emulated_latency_kernel_saturation(O, emu):
    rng = np.random.default_rng(1)
    for pat in (np.zeros(4 * 102, np.uint8), np.full(4 * 102, 255, np.uint8), (rng.integers(0, 2, 4 * 102) * 255).astype(np.uint8)):
        assert np.array_equal(emu.decode(pat, 96), O.decode_batch(96, pat)[0])
