"""Lane-level numpy emulation of vit_pk.hip's data layout (debug/validation tool).

Emulates one wave (4 frames): a pre-pass table of (M, 63-M) pairs (the kernel's first table
format; today it stores M only and splits the pre-pass differently, same values), the packed ACS
with the rotating lane<->state map, u16 biased arithmetic, renormalisation, the decision-history
layout, and a serial traceback through that layout with the physical position formula
l = ror5(state>>1, t mod 5).  Compares the decoded bytes with the oracle; also run by
tests/test_oracle_kat.py::test_packed_layout_emulation.  Run: python tests/tools/emulate_pk.py [framebits]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _vitpkg  # noqa: E402

O = _vitpkg.load_oracle()
M16 = 0xFFFF


def pk(f, a, b):
    lo = f(a & M16, b & M16)
    hi = f(a >> 16, b >> 16)
    return (lo & M16) | ((hi & M16) << 16)


add_sat = lambda a, b: pk(lambda x, y: np.minimum(x + y, M16), a, b)
sub_sat = lambda a, b: pk(lambda x, y: np.maximum(x.astype(np.int64) - y, 0).astype(np.uint32), a, b)
sub_wrap = lambda a, b: pk(lambda x, y: (x.astype(np.int64) - y) & M16, a, b)
pmin = lambda a, b: pk(np.minimum, a, b)
shr1 = lambda a: pk(lambda x, y: x >> 1, a, a)


def avg4(a, b):
    r = np.zeros_like(a)
    for k in range(4):
        x = (a >> (8 * k)) & 255
        y = (b >> (8 * k)) & 255
        r |= ((x + y + 1) >> 1) << (8 * k)
    return r


def perm(s0, s1, sel):
    s0 = np.broadcast_to(np.asarray(s0, np.uint32), (64,))
    s1 = np.broadcast_to(np.asarray(s1, np.uint32), (64,))
    sel = np.broadcast_to(np.asarray(sel, np.uint32), (64,))
    r = np.zeros(64, np.uint32)
    pool = [(s1 >> (8 * k)) & 255 for k in range(4)] + [(s0 >> (8 * k)) & 255 for k in range(4)]
    for k in range(4):
        c = (sel >> (8 * k)) & 255
        v = np.zeros(64, np.uint32)
        for j in range(8):
            v = np.where(c == j, pool[j], v)
        v = np.where(c >= 0x0D, 255, v)
        r |= v.astype(np.uint32) << (8 * k)
    return r


lane = np.arange(64, dtype=np.uint32)


def exchange(J, N0, N1):
    part = lane ^ (1 << J)
    bit = ((lane >> J) & 1).astype(bool)
    A = np.where(bit, N1[part], N0)
    B = np.where(bit, N1, N0[part])
    return A, B


def emulate(sym4, framebits):
    """sym4: (4, 4*(fb+6)) uint8 -> (4, fb//8) decoded bytes"""
    T = framebits + 6
    nblk = (T + 15) // 16
    l5, pair = lane & 31, lane >> 5
    toff = []
    for rho in range(5):
        i = ((l5 << rho) | (l5 >> (5 - rho))) & 31
        i0, i1, i2, i3, i4 = [(i >> k) & 1 for k in range(5)]
        c = (i1 ^ i2 ^ i4) | ((i0 ^ i1 ^ i2) << 1) | ((i0 ^ i3) << 2)
        toff.append(pair * 64 + c * 8)
    tau = lane >> 2
    pkf = ((lane >> 1) & 1) * 2 + (lane & 1)
    hb = np.where(tau & 1, 0x0C000C00, 0x0D000D00).astype(np.uint32)
    sel = [hb | np.uint32(0x00040000 + 0x00010001 * k) for k in range(4)]
    s32 = sym4.view(np.uint32).reshape(4, -1)
    A = np.where(l5 == 0, 0, 0x003F003F).astype(np.uint32)
    B = np.full(64, 0x003F003F, np.uint32)
    acc0 = np.zeros(64, np.uint32)
    acc1 = np.zeros(64, np.uint32)
    dec = np.zeros((nblk, 64, 2), np.uint32)
    for blk in range(nblk):
        t = blk * 16 + tau
        s = np.where(t < T, s32[pkf, np.minimum(t, T - 1)], 0).astype(np.uint32)
        r0 = perm(s, s, 0x00000000) ^ np.uint32(0xFF00FF00)
        r1 = perm(s, s, 0x01010101) ^ np.uint32(0xFFFF0000)
        r2 = perm(s, s, 0x02020202) ^ np.uint32(0xFFFF0000)
        r3 = perm(s, s, 0x03030303) ^ np.uint32(0xFF00FF00)
        P, Q = avg4(r0, r1), avg4(r2, r3)
        qlo, qhi = perm(Q, Q, 0x01000100), perm(Q, Q, 0x03020302)
        metlo = (avg4(P, qlo) >> 2) & np.uint32(0x3F3F3F3F)
        methi = (avg4(P, qhi) >> 2) & np.uint32(0x3F3F3F3F)
        mmlo, mmhi = np.uint32(0x3F3F3F3F) - metlo, np.uint32(0x3F3F3F3F) - methi
        h = (lane & 1).astype(bool)
        mine_met, mine_mm = np.where(h, methi, metlo), np.where(h, mmhi, mmlo)
        send_met, send_mm = np.where(h, metlo, methi), np.where(h, mmlo, mmhi)
        part_met, part_mm = send_met[lane ^ 1], send_mm[lane ^ 1]
        lo_met, hi_met = np.where(h, part_met, mine_met), np.where(h, mine_met, part_met)
        lo_mm, hi_mm = np.where(h, part_mm, mine_mm), np.where(h, mine_mm, part_mm)
        tab = np.zeros(2048 // 4, np.uint32)
        for k in range(4):
            tab[lane * 8 + 2 * k] = perm(hi_met, lo_met, sel[k])
            tab[lane * 8 + 2 * k + 1] = perm(hi_mm, lo_mm, sel[k])
        v = blk % 5
        for J in range(16):
            rho = (v + J) % 5
            off = (toff[rho] + J * 128) // 4
            Mv, MMv = tab[off], tab[off + 1]
            m0, m1, m2, m3 = add_sat(A, Mv), add_sat(B, MMv), add_sat(A, MMv), add_sat(B, Mv)
            n0, n1 = pmin(m0, m1), pmin(m2, m3)
            x01, x23 = sub_wrap(m0, m1), sub_wrap(m2, m3)
            acc0 = (x01 & np.uint32(0x80008000)) | shr1(acc0)
            acc1 = (x23 & np.uint32(0x80008000)) | shr1(acc1)
            if J & 1:
                z = n0[(lane >> 5) << 5]
                over = sub_sat(z, np.uint32(0xFF96FF96))
                K = pk(lambda x, y: np.minimum(x, 1) * 63 + 0xFF00, over, over)
                n0, n1 = sub_sat(n0, K), sub_sat(n1, K)
            A, B = exchange(4 - rho, n0, n1)
        dec[blk, :, 0], dec[blk, :, 1] = acc0, acc1
    # serial traceback through the layout (validates the position formula)
    out = np.zeros((4, framebits // 8), np.uint8)
    for fi in range(4):
        E = 0
        for n in range(framebits - 1, -1, -1):
            t = n + 6
            rho = t % 5
            vv, nb = (E >> 3) & 31, (E >> 2) & 1
            y = vv | (vv << 5)
            l = (y >> rho) & 31
            w = int(dec[t >> 4, (fi >> 1) * 32 + l, nb])
            k = ((w >> ((t & 15) + 16 * (fi & 1))) & 1) ^ 1
            E = ((E >> 1) | (k << 7)) & 0xFF
            if n % 8 == 0:
                out[fi, n >> 3] = E
    return out


if __name__ == "__main__":
    fb = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    sym = np.concatenate([O.noisy_frames(2, fb, seed=5), O.uniform_symbols(2 * O.sym_len(fb), seed=6).reshape(2, -1)])
    want = O.decode_batch(fb, sym)
    got = emulate(sym, fb)
    for f in range(4):
        nz = np.nonzero(got[f] != want[f])[0]
        print("frame", f, "OK" if nz.size == 0 else "MISMATCH first byte %d of %d (%d bad)" % (nz[0], fb // 8, nz.size))
