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
    """bench.py's choice: a bindable port BELOW the kernel's ephemeral range (a bind(0) port can be taken by any outgoing
    connection before the ranks listen on it)"""
    sys.path.insert(0, ROOT)
    import bench
    return bench._free_port()


def _worker(rank, world, port, nframes, framebits, q, pipeline=None):
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
                return torch.empty((0, (framebits + 7) // 8), dtype=torch.uint8)
            return torch.from_numpy(O.decode_batch(framebits, local.numpy()))

        def decode_into(local, out):
            out.copy_(decode(local))

        olen = (framebits + 7) // 8
        if pipeline is None:
            full = sharding.decode_sharded(sym_all, nframes, framebits, decode)
        else:
            chunk, rootf = pipeline
            full = torch.full((nframes, olen), 0xEE, dtype=torch.uint8) if rank == 0 else None
            for _ in range(2):  # twice: buffers and message order must survive a second pass
                plan = sharding.decode_stream(sym_all, full, nframes, framebits, decode_into, chunk, rootf)
            assert sum(plan.block(k, r)[1] for k in range(plan.nchunks) for r in range(world)) == nframes
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


def _run(world, nframes, framebits, pipeline):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, framebits, q, pipeline)) for r in range(world)]
    [p.start() for p in procs]
    status, equal, counts = q.get(timeout=180)
    [p.join(60) for p in procs]
    assert status == "ok", equal
    assert equal
    assert all(p.exitcode == 0 for p in procs)


def test_round_robin_partial_last_byte():
    """framebits % 8 != 0: (framebits+7)//8 output bytes per frame through scatter and gather"""
    _run(2, 9, 770, None)


@pytest.mark.parametrize("nframes,chunk,rootf", [(37, 4, None), (64, 8, 24), (5, 16, None), (33, 3, 1)])
def test_chunked_pipeline_scatter_decode_gather(nframes, chunk, rootf):
    """decode_stream: contiguous blocks, double-buffered receives, outputs sent back while the next chunk decodes"""
    _run(2, nframes, 288, (chunk, rootf))


def test_chunked_pipeline_three_ranks_partial_byte():
    _run(3, 29, 10, (2, 3))


def test_stream_plan_math():
    sys.path.insert(0, ROOT)
    import _vitpkg
    _vitpkg.load_package()
    from importlib import import_module
    sh = import_module("viterbi_dll_amd.sharding")
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            for chunk, rootf, root in ((1, None, 0), (4, None, 0), (16, 40, 0), (5, 2, w - 1)):
                plan = sh.StreamPlan(n, w, chunk, rootf, root)
                seen = []
                for k in range(plan.nchunks):
                    for r in range(w):
                        lo, c = plan.block(k, r)
                        assert 0 <= c <= plan.max_block(r)
                        seen += list(range(lo, lo + c))
                        assert all(plan.owner(f) == r for f in range(lo, lo + c))
                assert sorted(seen) == list(range(n))  # every frame exactly once


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
