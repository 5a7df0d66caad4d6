"""Frame sharding across the GPUs of one node (BASELINE config 4).

Frames are independent, so the decode itself needs no collective.  Two ways to feed N ranks:

* ``local shards`` (bench.py default): every rank owns its frames already; nothing moves.
* ``round-robin scatter/gather`` (this module): rank 0 holds the whole batch, frame f goes to
  rank f mod N, every rank decodes its share, rank 0 gathers the decoded bytes back
  into the original order.  With backend "nccl" this is RCCL over xGMI; the same code runs on
  gloo/CPU tensors, which is how tests/test_shard_gloo.py exercises it.

``decode`` is injected: on a GPU it is the HIP path (decode_batch_dev); the CPU tests inject the
oracle so that the plumbing can be checked without a device.
"""
import torch
import torch.distributed as dist


def shard_indices(nframes, rank, world):
    """Indices of the frames rank `rank` decodes: f with f % world == rank."""
    return torch.arange(rank, max(nframes, rank), world)


def shard_count(nframes, rank, world):
    return (nframes - rank + world - 1) // world if rank < nframes else 0


def scatter_frames(sym_all, nframes, sym_len, root=0, group=None):
    """Root passes sym_all [nframes, sym_len] uint8 (others: None).  Returns this rank's
    [count, sym_len] tensor on the same device type as the process group works with."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    count = shard_count(nframes, rank, world)
    maxc = shard_count(nframes, 0, world)
    dev = sym_all.device if sym_all is not None else _default_device()
    recv = torch.empty((maxc, sym_len), dtype=torch.uint8, device=dev)
    if rank == root:
        chunks = []
        for r in range(world):
            part = sym_all[r::world]
            if part.shape[0] < maxc:  # pad ragged tails so every rank receives the same shape
                pad = torch.zeros((maxc - part.shape[0], sym_len), dtype=torch.uint8, device=dev)
                part = torch.cat([part, pad])
            chunks.append(part.contiguous())
        dist.scatter(recv, chunks, src=root, group=group)
    else:
        dist.scatter(recv, None, src=root, group=group)
    return recv[:count]


def gather_outputs(out_local, nframes, out_len, root=0, group=None):
    """Inverse of scatter_frames for the decoded bytes; returns [nframes, out_len] on root."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    maxc = shard_count(nframes, 0, world)
    dev = out_local.device
    send = out_local
    if send.shape[0] < maxc:
        send = torch.cat([send, torch.zeros((maxc - send.shape[0], out_len), dtype=torch.uint8, device=dev)])
    send = send.contiguous()
    if rank == root:
        bufs = [torch.empty((maxc, out_len), dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(send, bufs, dst=root, group=group)
        full = torch.empty((nframes, out_len), dtype=torch.uint8, device=dev)
        for r in range(world):
            c = shard_count(nframes, r, world)
            full[r::world] = bufs[r][:c]
        return full
    dist.gather(send, None, dst=root, group=group)
    return None


def decode_sharded(sym_all, nframes, framebits, decode, root=0, group=None):
    """scatter -> decode(local_sym [n, sym_len]) -> [n, framebits//8] -> gather.  Root returns the
    decoded batch in the original frame order, other ranks None."""
    sym_len = 4 * (framebits + 6)
    local = scatter_frames(sym_all, nframes, sym_len, root, group)
    out_local = decode(local)
    return gather_outputs(out_local, nframes, framebits // 8, root, group)


def _default_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
