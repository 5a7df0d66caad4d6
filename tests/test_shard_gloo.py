"""CPU, world_size 2, gloo: the multi-GPU plumbing (round-robin scatter, per-rank decode,
gather) must reproduce the unsharded result.  The per-rank decoder is injected; here it is the
oracle (test infrastructure), on a GPU box bench.py injects the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, framebits, q):
    sys.path.insert(0, ROOT)
    import _vitpkg
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        V = _vitpkg.load_package()
        O = _vitpkg.load_oracle()
        from importlib import import_module
        sharding = import_module("viterbi_dll_amd.sharding")
        sym_all = None
        if rank == 0:
            nu = nframes // 2
            sym = np.concatenate([O.noisy_frames(nframes - nu, framebits, seed=5),
                                  O.uniform_symbols(nu * O.sym_len(framebits), seed=6).reshape(nu, O.sym_len(framebits))])
            sym_all = torch.from_numpy(sym)

        def decode(local):  # oracle stands in for the HIP path on CPU
            if local.shape[0] == 0:
                return torch.empty((0, framebits // 8), dtype=torch.uint8)
            return torch.from_numpy(O.decode_batch(framebits, local.numpy()))

        full = sharding.decode_sharded(sym_all, nframes, framebits, decode)
        if rank == 0:
            want = O.decode_batch(framebits, sym_all.numpy())
            q.put(("ok", bool(np.array_equal(full.numpy(), want)),
                   [int(sharding.shard_count(nframes, r, world)) for r in range(world)]))
        assert V.EXPORTS  # package imports without a GPU
    except Exception as e:  # pragma: no cover
        q.put(("err", repr(e), None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nframes", [37, 8, 1])
def test_round_robin_scatter_decode_gather(nframes):
    world, framebits = 2, 288
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, framebits, q)) for r in range(world)]
    [p.start() for p in procs]
    status, equal, counts = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert status == "ok", equal
    assert equal
    assert sum(counts) == nframes and counts[0] - counts[1] in (0, 1)
    assert all(p.exitcode == 0 for p in procs)


def test_shard_index_math():
    sys.path.insert(0, ROOT)
    import _vitpkg
    _vitpkg.load_package()
    from importlib import import_module
    sh = import_module("viterbi_dll_amd.sharding")
    for n in (0, 1, 7, 8, 65536):
        for w in (1, 2, 4, 8):
            idx = [sh.shard_indices(n, r, w) for r in range(w)]
            assert sorted(torch.cat(idx).tolist()) == list(range(n))
            assert [len(i) for i in idx] == [sh.shard_count(n, r, w) for r in range(w)]
