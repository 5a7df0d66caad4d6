"""Lane-level numpy emulation of vit_pk8.hip's data layout (debug/validation tool, like emulate_pk.py).

EIGHT frames per wavefront: four rows of 16 lanes, a row = a frame pair (the two 16-bit halves of a VGPR), and per
lane FOUR metric registers N[x][y].  A trellis state's 6 bits sit in 4 lane bits (L3..L0) and 2 register bits (X, Y).
Time runs in cycles of 5 steps with ONE active register bit R (X in even cycles, Y in odd ones):
    step p = 0      : butterfly on R                      (R's bit has aged to s5 while R was dormant: nothing moves)
    step p = 1 .. 4 : butterfly on R after the post-swap of step p-1 exchanged R with lane bit J = 4 - p
so four of five steps move data (one reg-bit <-> lane-bit transposition of both register pairs) and one moves none.
State 0 is always lane 0 of its row, register N[0][0].

Emulates the packed u16 arithmetic (biased / 0-based alternation, renormalisation), the table classes per lane and
phase, the decision-history words and a serial traceback in physical coordinates; compares with the oracle.
Run: python tests/tools/emulate_pk8.py [framebits]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _vitpkg  # noqa: E402

O = _vitpkg.load_oracle()
M16 = 0xFFFF
lane = np.arange(64, dtype=np.uint32)
L4, ROW = lane & 15, lane >> 4
X, Y = 4, 5  # position numbers of the two register bits (0..3 = lane bits)


def pk(f, a, b):
    lo = f(a & M16, b & M16)
    hi = f(a >> 16, b >> 16)
    return ((lo & M16) | ((hi & M16) << 16)).astype(np.uint32)


add_sat = lambda a, b: pk(lambda x, y: np.minimum(x.astype(np.int64) + y, M16), a, b)
sub_sat = lambda a, b: pk(lambda x, y: np.maximum(x.astype(np.int64) - y, 0), a, b)
sub_wrap = lambda a, b: pk(lambda x, y: (x.astype(np.int64) - y) & M16, a, b)
pmin = lambda a, b: pk(np.minimum, a, b)


def schedule(t):
    """(active register position, lane bit swapped AFTER the butterfly of step t or None)"""
    c, p = divmod(t, 5)
    R = X if c % 2 == 0 else Y
    return R, (3 - p if p < 4 else None)


def bit_positions(T):
    """pos[t][k] = position (0..3 lane bit, 4 = X, 5 = Y) of state bit k BEFORE step t"""
    pos = {5: X, 0: Y, 4: 3, 3: 2, 2: 1, 1: 0}
    out = []
    for t in range(T + 1):
        out.append(dict(pos))
        R, J = schedule(t)
        assert pos[5] == R, (t, pos)
        new = {0: R}
        for k in range(5):
            new[k + 1] = pos[k]
        if J is not None:  # post-swap: the contents of positions R and J change places
            inv = {v: k for k, v in new.items()}
            kR, kJ = inv[R], inv[J]
            new[kR], new[kJ] = J, R
        pos = new
    return out


def classes(pos_t, R):
    """table class of the butterfly of every lane for other-register-bit o = 0, 1 at a step with map pos_t"""
    other = Y if R == X else X
    res = []
    for o in (0, 1):
        ib = []
        for k in range(5):
            p = pos_t[k]
            ib.append(((L4 >> p) & 1) if p < 4 else np.full(64, o, np.uint32) if p == other else None)
        assert all(b is not None for b in ib)
        i0, i1, i2, i3, i4 = ib
        res.append((i1 ^ i2 ^ i4) | ((i0 ^ i1 ^ i2) << 1) | ((i0 ^ i3) << 2))
    return res


def metric8(s4):
    """the 8 pavgb-tree metrics of one frame-step; s4 = its 4 symbols"""
    avg = lambda a, b: (a + b + 1) >> 1
    out = []
    for c in range(8):
        b0, b1, b2 = c & 1, (c >> 1) & 1, (c >> 2) & 1
        x = [int(s4[0]) ^ (255 * b0), int(s4[1]) ^ (255 * b1), int(s4[2]) ^ (255 * b2), int(s4[3]) ^ (255 * b0)]
        out.append(avg(avg(x[0], x[1]), avg(x[2], x[3])) >> 2)
    return out


def emulate(sym8, framebits, ge=False):
    """sym8: (8, 4*(fb+6)) uint8 -> (8, (fb+7)//8) decoded bytes"""
    T = framebits + 6
    nblk = (T + 15) // 16
    pos = bit_positions(nblk * 16)
    N = [[np.where(L4 == 0, 0, 0x003F003F).astype(np.uint32), np.full(64, 0x003F003F, np.uint32)],
         [np.full(64, 0x003F003F, np.uint32), np.full(64, 0x003F003F, np.uint32)]]  # N[x][y]
    hist = np.zeros((nblk, 64, 2, 2), np.uint32)  # [block][lane][x][y], step j at bit j of each half
    rc = np.uint32(0x8069806A if ge else 0x80688069)
    for t in range(nblk * 16):
        R, J = schedule(t)
        odd = t & 1
        # table of this step: M per (row, class): lo half = frame 2*row, hi half = frame 2*row + 1; even steps re-bias
        tab = np.zeros((4, 8), np.uint32)
        for g in range(4):
            ma = metric8(sym8[2 * g, 4 * t:4 * t + 4]) if t < T else [0] * 8
            mb = metric8(sym8[2 * g + 1, 4 * t:4 * t + 4]) if t < T else [0] * 8
            for c in range(8):
                bias = 0 if odd else 0xFF00
                tab[g, c] = (ma[c] + bias) | ((mb[c] + bias) << 16)
        cls = classes(pos[t], R)
        for o in (0, 1):
            a, b = (N[0][o], N[1][o]) if R == X else (N[o][0], N[o][1])
            Mv = tab[ROW, cls[o]]
            MMv = (np.uint32(0x003F003F) if odd else np.uint32(0xFE40FE3F)) - Mv
            m0, m1, m2, m3 = add_sat(a, Mv), add_sat(b, MMv), add_sat(a, MMv), add_sat(b, Mv)
            n0, n1 = pmin(m0, m1), pmin(m2, m3)
            s01 = (sub_wrap(m0, m1) >> 15) & np.uint32(0x00010001)  # sign = NOT decision
            s23 = (sub_wrap(m2, m3) >> 15) & np.uint32(0x00010001)
            sl0, sl1 = ((0, o), (1, o)) if R == X else ((o, 0), (o, 1))
            hist[t >> 4, :, sl0[0], sl0[1]] |= s01 << (t & 15)
            hist[t >> 4, :, sl1[0], sl1[1]] |= s23 << (t & 15)
            N[sl0[0]][sl0[1]], N[sl1[0]][sl1[1]] = n0, n1
        if odd:
            z = N[0][0][ROW << 4]  # state 0: lane 0 of the row
            w = (z.astype(np.uint64) + rc).astype(np.uint32)
            K = ((w >> 15) & np.uint32(0x00010001)) * np.uint32(63) + np.uint32(0xFF00FF00)
            for x in (0, 1):
                for y in (0, 1):
                    N[x][y] = sub_sat(N[x][y], K)
        if J is not None:  # transposition of register bit R with lane bit J, for both values of the other bit
            part, hi = lane ^ (1 << J), ((lane >> J) & 1).astype(bool)
            for o in (0, 1):
                n0, n1 = (N[0][o], N[1][o]) if R == X else (N[o][0], N[o][1])
                A = np.where(hi, n1[part], n0)
                B = np.where(hi, n1, n0[part])
                if R == X:
                    N[0][o], N[1][o] = A, B
                else:
                    N[o][0], N[o][1] = A, B
    # serial traceback in physical coordinates
    out = np.zeros((8, (framebits + 7) // 8), np.uint8)
    for fi in range(8):
        g, h = fi >> 1, fi & 1
        l4 = x = y = 0  # state 0 after the last step
        bits = np.zeros(((framebits + 7) // 8) * 8, np.uint8)
        for t in range(T - 1, 5, -1):
            R, J = schedule(t)
            if J is not None:  # undo the post-swap of step t
                r = x if R == X else y
                lj = (l4 >> J) & 1
                l4 = (l4 & ~(1 << J)) | (r << J)
                if R == X:
                    x = lj
                else:
                    y = lj
            k = 1 - ((int(hist[t >> 4, g * 16 + l4, x, y]) >> ((t & 15) + 16 * h)) & 1)
            if R == X:
                x = k
            else:
                y = k
            bits[t - 6] = k
        out[fi] = np.packbits(bits)
    return out


if __name__ == "__main__":
    fb = int(sys.argv[1]) if len(sys.argv) > 1 else 136
    sym = np.concatenate([O.noisy_frames(3, fb, seed=5), O.uniform_symbols(3 * O.sym_len(fb), seed=6).reshape(3, -1),
                          O.hard_random_symbols(2, fb, seed=7)])
    for ge in (False, True):
        want = O.decode_batch(fb, sym, ge=ge)
        got = emulate(sym, fb, ge)
        for f in range(8):
            nz = np.nonzero(got[f] != want[f])[0]
            print("ge", ge, "frame", f, "OK" if nz.size == 0 else "MISMATCH first byte %d of %d (%d bad)" % (nz[0], fb // 8, nz.size))
