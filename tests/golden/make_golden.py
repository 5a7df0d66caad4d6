"""Generates tests/golden/golden.json: regression vectors for the hot path.

PROVENANCE: produced by THIS repo's oracle (oracle/vit_oracle.c), not by the reference --
the reference cannot be built or run under this project's rules (it needs stand-ins for
<windows.h>/<psapi.h> and MASM data).  The oracle itself is pinned by SURVEY 8c's KATs
(tests/test_oracle_kat.py).  Inputs AND expected outputs are stored (inputs zlib-compressed + base64, plus
their FNV-1a-64), so that the GPU suite can compare the HIP path with the committed bytes without loading
the oracle at all.

Decode cases carry TWO expected outputs: `out_hex` for the `> 150` renormalise comparator (the reference's
C decoders, deconvolve.cpp:407-412) and `out_ge_hex` for `>= 150` (its MASM decoders, decon_avx2.asm:94-118);
on soft-decision input they are equal, the hard-decision cases are picked so that they DIFFER.  The ge
outputs come from the oracle's restatement of asm text that cannot be assembled here (parity unpinned).
"""
import base64
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg  # noqa: E402

O = _vitpkg.load_oracle()
KAT = 88172645463325252


def zb64(a):
    return base64.b64encode(zlib.compress(np.ascontiguousarray(a, np.uint8).tobytes(), 9)).decode()


def make_sym(fb, kind, seed):
    if kind == "uniform":
        return O.uniform_symbols(O.sym_len(fb), seed=seed)
    if kind == "noisy":
        return O.noisy_frames(1, fb, seed=seed)[0]
    if kind == "hard_random":
        return O.hard_random_symbols(1, fb, seed=seed)[0]
    if kind == "hard_flipped":
        return O.hard_flipped_frames(1, fb, flip=0.2, seed=seed)[0]
    raise ValueError(kind)


g = {"provenance": "oracle/vit_oracle.c (own restatement); see make_golden.py", "decode": [], "rs": []}
cases = [(768, "uniform", KAT), (288, "uniform", KAT), (768, "noisy", 11), (1536, "noisy", 12), (96, "uniform", 13),
         (8, "uniform", 14), (3072, "noisy", 15), (2304, "uniform", 16),
         # round 3: the longest DAB sub-channel frame, the ABI's maximum, a partial last byte, the shortest frame
         (6912, "uniform", 17), (9216, "noisy", 18), (770, "uniform", 19), (2, "uniform", 20)]
for fb, kind, seed in cases:
    sym = make_sym(fb, kind, seed)
    out, out_ge = O.decode_batch(fb, sym)[0], O.decode_batch(fb, sym, ge=True)[0]
    g["decode"].append({"framebits": fb, "kind": kind, "seed": seed, "sym_fnv1a64": "%016x" % O.fnv1a64(sym),
                        "sym_zb64": zb64(sym), "out_hex": out.tobytes().hex(), "out_ge_hex": out_ge.tobytes().hex()})
# hard-decision cases on which the two comparators differ: first seed that separates them
for fb, kind in [(768, "hard_random"), (768, "hard_flipped"), (3072, "hard_flipped"), (6912, "hard_flipped"),
                 (3072, "hard_random")]:
    for seed in range(100, 400):
        sym = make_sym(fb, kind, seed)
        out, out_ge = O.decode_batch(fb, sym)[0], O.decode_batch(fb, sym, ge=True)[0]
        if not np.array_equal(out, out_ge):
            break
    else:
        raise SystemExit("no separating seed for %s %d" % (kind, fb))
    g["decode"].append({"framebits": fb, "kind": kind, "seed": seed, "sym_fnv1a64": "%016x" % O.fnv1a64(sym),
                        "sym_zb64": zb64(sym), "out_hex": out.tobytes().hex(), "out_ge_hex": out_ge.tobytes().hex()})

# ---- RS(120,110) ----
ato, iof = O.rs_tables()


def gmul(a, b):
    return 0 if a == 0 or b == 0 else int(ato[int(iof[a]) + int(iof[b])])


GEN = [1]
for i in range(10):  # g(x) = prod (x + alpha^i), lowest coefficient first
    ng = [0] * (len(GEN) + 1)
    for j, c in enumerate(GEN):
        ng[j + 1] ^= c
        ng[j] ^= gmul(c, int(ato[i]))
    GEN = ng


def padding_parity(pos, val):
    """parity bytes (highest degree first) of the FULL-LENGTH RS(255,245) codeword whose only non-zero data symbol is
    `val` at position `pos` (0 = x^254) inside the 135 virtual padding symbols: remainder of val * x^(254-pos) mod g."""
    r = [0] * 10  # r[9] = highest coefficient
    for k in range(pos, 245):  # feed the data symbols pos..244 (all zero but the first), Horner / LFSR
        d = val if k == pos else 0
        fb = d ^ r[9]
        r = [gmul(fb, GEN[0])] + [r[i - 1] ^ gmul(fb, GEN[i]) for i in range(1, 10)]
    return np.array(r[::-1], np.uint8)


rng = np.random.default_rng(2024)


def rs_case(rsdims, errs, note, pad_cols=()):
    p = np.empty((120, rsdims), np.uint8)
    for j, ne in enumerate(errs):
        cw = O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8))
        if j in pad_cols:
            # the received word is ONE symbol away from a full-length codeword - in the virtual padding: the locator's
            # root lies below PAD + 1, Forney skips it silently, it still counts (rschecksf.cpp:348)
            cw[110:] ^= padding_parity(int(rng.integers(0, 135)), int(rng.integers(1, 256)))
        else:
            pos = rng.choice(120, ne, replace=False)
            cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
        p[:, j] = cw
    p = p.reshape(-1)
    ret, out = O.rs_check_superframe(p, rsdims, np.full(110 * rsdims, 0xA5, np.uint8))
    g["rs"].append({"rsdims": rsdims, "errors": errs, "note": note, "pad_cols": list(pad_cols), "p_hex": p.tobytes().hex(),
                    "ret": int(ret), "out_hex": out.tobytes().hex()})
    return int(ret), out


rs_case(4, [0, 2, 5, 1], "corrected")
rs_case(4, [1, 6, 0, 0], "fails at column 1")
rs_case(6, [5, 5, 5, 5, 5, 5], "five errors in every column")
rs_case(3, [0, 0, 7], "fails at the last column")
# round 3: the RSDims of BASELINE config 5 (24) and 16
assert rs_case(24, [0] * 24, "clean")[0] == 0
assert rs_case(24, [int(x) for x in rng.integers(0, 6, 24)], "corrected, 0..5 errors per column")[0] > 0
e = [int(x) for x in rng.integers(0, 4, 24)]
e[11] = 7
e[17] = 6
ret, out = rs_case(24, e, "first failure in the middle (column 11): columns 11..23 stay untouched")
assert ret == -1 and (out.reshape(110, 24)[:, 11:] == 0xA5).all() and not (out.reshape(110, 24)[:, :11] == 0xA5).all()
e = [int(x) for x in rng.integers(0, 3, 16)]
ret, out = rs_case(16, e, "column 5 decodes to a root in the virtual padding: counted, nothing patched", pad_cols=(5,))
assert ret == sum(e) - e[5] + 1, (ret, e)
e = [0, 1, 0, 2, 0, 0, 3, 0, 0, 0, 1, 0, 0, 0, 0, 9]
assert rs_case(16, e, "clean and corrected columns, failure at the last column")[0] == -1
assert rs_case(16, [0] * 16, "clean")[0] == 0

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json"), "w") as f:
    json.dump(g, f, indent=1)
print("wrote golden.json:", len(g["decode"]), "decode +", len(g["rs"]), "rs cases")
