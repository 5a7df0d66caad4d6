"""Generates tests/golden/golden.json: regression vectors for the hot path.

PROVENANCE: produced by THIS repo's oracle (oracle/vit_oracle.c), not by the reference --
the reference cannot be built or run under this project's rules (it needs stand-ins for
<windows.h>/<psapi.h> and MASM data).  The oracle itself is pinned by SURVEY 8c's KATs
(tests/test_oracle_kat.py).  Inputs AND expected outputs are stored (inputs as base64, plus their seed and FNV-1a-64), so that the GPU
suite can compare the HIP path with the committed bytes without loading the oracle at all.
"""
import base64
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg  # noqa: E402

O = _vitpkg.load_oracle()
g = {"provenance": "oracle/vit_oracle.c (own restatement); see make_golden.py", "decode": [], "rs": []}
for fb, kind, seed in [(768, "uniform", 88172645463325252), (288, "uniform", 88172645463325252),
                       (768, "noisy", 11), (1536, "noisy", 12), (96, "uniform", 13), (8, "uniform", 14),
                       (3072, "noisy", 15), (2304, "uniform", 16)]:
    sym = O.uniform_symbols(O.sym_len(fb), seed=seed) if kind == "uniform" else O.noisy_frames(1, fb, seed=seed)[0]
    out = O.decode_batch(fb, sym)[0]
    g["decode"].append({"framebits": fb, "kind": kind, "seed": seed, "sym_fnv1a64": "%016x" % O.fnv1a64(sym),
                        "sym_b64": base64.b64encode(sym.tobytes()).decode(),
                        "out_hex": out.tobytes().hex()})
rng = np.random.default_rng(2024)
for rsdims, errs in [(4, [0, 2, 5, 1]), (4, [1, 6, 0, 0]), (6, [5, 5, 5, 5, 5, 5]), (3, [0, 0, 7])]:
    p = np.empty((120, rsdims), np.uint8)
    for j, ne in enumerate(errs):
        cw = O.rs_encode(rng.integers(0, 256, 110, dtype=np.uint8))
        pos = rng.choice(120, ne, replace=False)
        cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
        p[:, j] = cw
    p = p.reshape(-1)
    ret, out = O.rs_check_superframe(p, rsdims, np.full(110 * rsdims, 0xA5, np.uint8))
    g["rs"].append({"rsdims": rsdims, "errors": errs, "p_hex": p.tobytes().hex(), "ret": int(ret),
                    "out_hex": out.tobytes().hex()})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json"), "w") as f:
    json.dump(g, f, indent=1)
print("wrote golden.json:", len(g["decode"]), "decode +", len(g["rs"]), "rs cases")
