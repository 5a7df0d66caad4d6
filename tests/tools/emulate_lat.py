#!/usr/bin/env python3
"""CPU emulation of vit_lat.hip's ALGORITHM (rotating lane <-> state map, partner fetch, class tables, decision
masks, per-lane history words, blocked speculative traceback) with numpy vectors standing in for the 64 lanes.
Checks the design against the oracle without a GPU: python tests/tools/emulate_lat.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg  # noqa: E402

O = _vitpkg.load_oracle()
LANES = np.arange(64)


def avg(a, b):
    return (a.astype(np.int32) + b + 1) >> 1


def class_metrics(s4):
    """8 metrics of one step: class c = b0 | b1<<1 | b2<<2, masks (b0,b1,b2,b0) on the 4 symbols"""
    out = np.zeros(8, np.int32)
    for c in range(8):
        b = [c & 1, (c >> 1) & 1, (c >> 2) & 1, c & 1]
        x = [int(s4[k]) ^ (255 if b[k] else 0) for k in range(4)]
        out[c] = avg(np.int32(avg(np.int32(x[0]), x[1])), avg(np.int32(x[2]), x[3])) >> 2
    return out & 63


def lane_class(rho):
    i = np.zeros(64, np.int64)
    for k in range(5):
        i |= ((LANES >> ((k - rho) % 6)) & 1) << k
    i0, i1, i2, i3, i4 = [(i >> k) & 1 for k in range(5)]
    return (i1 ^ i2 ^ i4) | ((i0 ^ i1 ^ i2) << 1) | ((i0 ^ i3) << 2)


def decode(sym, fb):
    T = fb + 6
    s = sym.reshape(T, 4)
    m = np.where(LANES == 0, 0, 63).astype(np.int64)
    dec = np.zeros((T, 64), np.int64)  # decision per step and lane (the kernel packs 32 steps per word)
    cls = [lane_class(r) for r in range(6)]
    for t in range(T):
        rho = t % 6
        j = (5 - rho) % 6
        M = class_metrics(s[t])[cls[rho]]
        p = m[LANES ^ (1 << j)]
        om = np.minimum(m + M, 255)
        pm = np.minimum(p + 63 - M, 255)
        n = np.minimum(om, pm)
        lj = (LANES >> j) & 1
        hi = np.where(lj == 1, om, pm)       # the candidate that came from predecessor i+32
        dec[t] = n == hi                     # tie -> 1 on both sides (deconvolve.cpp:352-374)
        assert np.array_equal(dec[t].astype(bool), np.where(lj == 1, om <= pm, pm <= om))
        if t & 1 and n[0] > 150:
            n = np.maximum(n - 63, 0)
        m = n
    # traceback, blocked + speculative like the kernel
    BL = 6 * ((fb + 64 * 6 - 1) // (64 * 6))
    WARM = 30
    bits = np.zeros(fb, np.uint8)
    L_out = np.zeros(64, np.int64)
    L_in = np.zeros(64, np.int64)
    blockbits = {}

    def trace(q, L, i_from, i_to, i_max, record):
        tbase = 6 + q * BL
        for ii in range(i_from, i_to - 1, -1):
            if ii > i_max:
                continue
            t = tbase + ii
            d = int(dec[t, L])
            jj = (5 - t) % 6
            assert jj == (5 - ii) % 6
            L = (L & ~(1 << jj)) | (d << jj)
            if record:
                blockbits[(q, ii)] = d
        return L

    q_top = (T - 1 - 6) // BL
    info = []
    for q in range(64):
        tbase = 6 + q * BL
        has = tbase < T
        i_last = T - 1 - tbase if has else 0
        i_warm = BL - 1 + WARM
        i_start = min(i_last, i_warm)
        fixed = has and i_last <= i_warm
        info.append((has, i_last, i_start, fixed))
        if has:
            L = trace(q, 0, i_warm, BL, i_start, False)
            L_in[q] = L
            L_out[q] = trace(q, L, BL - 1, 0, i_start, True)
    for _ in range(65):
        changed = False
        new_in = [L_out[q + 1] if q < q_top else 0 for q in range(64)]
        for q in range(64):
            has, i_last, i_start, fixed = info[q]
            if has and not fixed and new_in[q] != L_in[q]:
                changed = True
                L_in[q] = new_in[q]
                L_out[q] = trace(q, int(new_in[q]), BL - 1, 0, BL - 1, True)
        if not changed:
            break
    for (q, ii), d in blockbits.items():
        b = q * BL + ii
        if b < fb and 6 + b < T:
            bits[b] = d
    nbytes = (fb + 7) // 8
    padded = np.zeros(nbytes * 8, np.uint8)
    padded[:fb] = bits
    return np.packbits(padded)


if __name__ == "__main__":
    bad = 0
    for fb, seed in ((768, 1), (288, 2), (10, 3), (770, 4), (96, 5), (1536, 6), (2, 7)):
        for kind in ("noisy", "uniform"):
            sym = (O.noisy_frames(1, fb, seed=seed)[0] if kind == "noisy"
                   else O.uniform_symbols(O.sym_len(fb), seed=seed))
            want = O.decode_batch(fb, sym)[0]
            got = decode(sym, fb)
            ok = np.array_equal(got, want)
            bad += not ok
            print(fb, kind, "ok" if ok else "MISMATCH")
    # saturation / renorm stress
    rng = np.random.default_rng(0)
    for pat in (np.zeros(3096, np.uint8), np.full(3096, 255, np.uint8), (rng.integers(0, 2, 3096) * 255).astype(np.uint8)):
        ok = np.array_equal(decode(pat, 768), O.decode_batch(768, pat)[0])
        bad += not ok
        print("stress", "ok" if ok else "MISMATCH")
    sys.exit(1 if bad else 0)
