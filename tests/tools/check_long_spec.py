#!/usr/bin/env python3
"""Long-frame kernel (csrc/vit_pk.hip, in-flight speculative parts + write-only spill) against the oracle over the input
families that exercise each of its paths: Eb/N0 = 3 dB (parts verified, a few fail and come back from the spill), 1.5 dB
(more part-level failures, waves stay speculative), 0 dB and uniform random bytes (waves give up tracing in flight), hard
decisions.  Run it against the product library and against the test builds
    VITERBI_AMD_LIB=tools/exp/libviterbi_sabotage.so   (-DVIT_SPEC_SABOTAGE: EVERY in-flight part fails its check)
    VITERBI_AMD_LIB=tools/exp/libviterbi_noinflight.so (-DVIT_LONG_INFLIGHT=0: nothing traced in flight)
usage: check_long_spec.py [frames per case]   -> one line per case, exit 1 on any difference."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
V.initialize()
V.set_renorm_ge(0)
V.set_kernel(2)
bad = 0
for fb in [784, 800, 1008, 1024, 1040, 1280, 1536, 2304, 3072, 4608, 6912, 9216, 9200]:
    fams = {
        "3dB": O.noisy_frames(n, fb, seed=fb),
        "1.5dB": O.noisy_frames(n, fb, seed=fb + 1, ebn0_db=1.5),
        "0dB": O.noisy_frames(n, fb, seed=fb + 2, ebn0_db=0.0),
        "random": O.uniform_symbols(n * O.sym_len(fb), seed=fb + 3).reshape(n, -1),
        "hard": O.hard_flipped_frames(n, fb, flip=0.1, seed=fb + 4),
    }
    for name, sym in fams.items():
        want = O.decode_batch(fb, sym, nthreads=8)
        d_out = torch.full((n, fb // 8), 0xEE, dtype=torch.uint8, device="cuda")
        V.decode_batch_dev(torch.from_numpy(np.ascontiguousarray(sym)).cuda(), d_out, fb, n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        nd = int((got != want).any(axis=1).sum())
        bad += nd
        print("framebits %5d %-6s frames %d differing %d%s" % (fb, name, n, nd, "" if nd == 0 else "  <-- first bad frame %d byte %d" % (
            int(np.argmax((got != want).any(axis=1))), int(np.argmax((got != want)[np.argmax((got != want).any(axis=1))])))), flush=True)
print("TOTAL differing frames:", bad)
sys.exit(1 if bad else 0)
