#!/usr/bin/env python3
"""What the long-frame kernel's fast groups do per input family (needs a -DVIT_DIAG_SPEC build in VITERBI_AMD_LIB):
groups, parts traced in flight, groups that gave up tracing in flight, parts traced after the forward pass (beyond part 0),
and how many of those had been traced in flight and failed their check.  Every output is compared with the oracle's decode of the
distinct frames the batch is tiled from.  usage: spec_stats.py [framebits] [frames]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _vitpkg  # noqa: E402

V = _vitpkg.load_package()
O = _vitpkg.load_oracle()
O.build()
import torch  # noqa: E402

fb = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
V.initialize()
V.set_renorm_ge(0)
V.set_kernel(2)
lib = V.lib()
distinct = 512
for name, mk in (("3dB", lambda: O.noisy_frames(distinct, fb, seed=11)),
                 ("2dB", lambda: O.noisy_frames(distinct, fb, seed=12, ebn0_db=2.0)),
                 ("1dB", lambda: O.noisy_frames(distinct, fb, seed=13, ebn0_db=1.0)),
                 ("0dB", lambda: O.noisy_frames(distinct, fb, seed=14, ebn0_db=0.0)),
                 ("random", lambda: O.uniform_symbols(distinct * O.sym_len(fb), seed=15).reshape(distinct, -1))):
    sym = mk()
    want = torch.from_numpy(O.decode_batch(fb, sym, nthreads=8)).cuda()
    d_sym = torch.from_numpy(sym).cuda().repeat(n // distinct, 1)
    d_out = torch.zeros((n, fb // 8), dtype=torch.uint8, device="cuda")
    lib.vit_diag_spec(None, 1)
    V.decode_batch_dev(d_sym, d_out, fb, n)
    torch.cuda.synchronize()
    c = np.zeros(8, np.uint64)
    lib.vit_diag_spec(c.ctypes.data_as(ctypes.c_void_p), 0)
    ok = bool((d_out.view(n // distinct, distinct, -1) == want.unsqueeze(0)).all())
    g = max(int(c[0]), 1)
    print(json.dumps({"framebits": fb, "frames": n, "input": name, "bit_exact": ok, "groups": int(c[0]), "parts_in_flight": int(c[1]),
                      "groups_gave_up": int(c[2]), "parts_after_forward": int(c[3]), "of_those_failed_check": int(c[4]),
                      "parts_per_group": (fb + 255) // 256, "failed_per_part_in_flight": round(int(c[4]) / max(int(c[1]), 1), 5)}), flush=True)
