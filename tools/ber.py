#!/usr/bin/env python3
"""BER/FER harness, the counterpart of viterbi-benchmark.cpp:293-329: random bits -> DAB mother
code -> AWGN at Eb/N0 -> decode on the GPU -> bit/frame error rates.  Frames are generated on the
device (bench.make_frames with return_bits).  usage: tools/ber.py [ebn0_db] [frames] [framebits]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _vitpkg  # noqa: E402
import bench  # noqa: E402


def run(ebn0_db=3.0, frames=5000, framebits=3072, seed=0, device="cuda:0"):
    V = _vitpkg.load_package()
    V.initialize()
    dev = torch.device(device)
    sym, bits = bench.make_frames(frames, framebits, seed=seed, device=dev, ebn0_db=ebn0_db, return_bits=True)
    out = torch.zeros((frames, framebits // 8), dtype=torch.uint8, device=dev)
    V.decode_batch_dev(sym, out, framebits, frames)
    torch.cuda.synchronize()
    dec = np.unpackbits(out.cpu().numpy(), axis=1)
    err = dec != bits.cpu().numpy().astype(np.uint8)
    return {"ebn0_db": ebn0_db, "frames": frames, "framebits": framebits, "bit_errors": int(err.sum()),
            "ber": float(err.mean()), "bad_frames": int(err.any(axis=1).sum()), "fer": float(err.any(axis=1).mean())}


if __name__ == "__main__":
    a = sys.argv[1:]
    print(json.dumps(run(float(a[0]) if a else 3.0, int(a[1]) if len(a) > 1 else 5000, int(a[2]) if len(a) > 2 else 3072)))
