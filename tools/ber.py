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
    if a and a[0] == "curve":  # BER/FER vs Eb/N0, 200k frames of 768 bits per point (error counts done on the device)
        for db in (0.0, 1.0, 2.0, 3.0, 4.0, 5.0):
            V = _vitpkg.load_package(); V.initialize()
            dev = torch.device("cuda:0")
            nerr = nfr = nbad = 0
            for chunk in range(4):
                sym, bits = bench.make_frames(50000, 768, seed=100 * int(db) + chunk, device=dev, ebn0_db=db, return_bits=True)
                out = torch.zeros((50000, 96), dtype=torch.uint8, device=dev)
                V.decode_batch_dev(sym, out, 768, 50000)
                w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.int32, device=dev)
                ref = (bits.view(50000, 96, 8) * w).sum(dim=2).to(torch.uint8)
                x = (out ^ ref).to(torch.int32)
                pop = sum(((x >> k) & 1) for k in range(8))
                nerr += int(pop.sum()); nbad += int((x != 0).any(dim=1).sum()); nfr += 50000
            print(json.dumps({"ebn0_db": db, "frames": nfr, "framebits": 768, "bit_errors": nerr, "ber": nerr / (nfr * 768.0),
                              "bad_frames": nbad, "fer": nbad / float(nfr)}), flush=True)
        sys.exit(0)
    print(json.dumps(run(float(a[0]) if a else 3.0, int(a[1]) if len(a) > 1 else 5000, int(a[2]) if len(a) > 2 else 3072)))
