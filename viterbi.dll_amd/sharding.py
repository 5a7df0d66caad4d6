"""Frame sharding across the GPUs of one node (BASELINE config 4).

Frames are independent, so the decode itself needs no collective.  Three ways to feed N ranks:

* ``local shards`` (bench.py default): every rank owns its frames already; nothing moves.
* ``decode_sharded`` - the plain round-robin form of SURVEY 8e: rank 0 holds the whole batch, frame f
  goes to rank f mod N with ONE scatter, every rank decodes its share, ONE gather puts the decoded
  bytes back into the original order.  Nothing overlaps; kept as the simple reference form.
* ``decode_stream`` - the chunked, overlapped pipeline: the stream is cut into chunks of
  ``root_frames + (N-1) * chunk_frames`` consecutive frames; inside a chunk the root keeps the first
  ``root_frames`` and peer r takes the r-th block of ``chunk_frames`` frames (round-robin at block
  granularity: every slice is CONTIGUOUS in the root's buffers, so it is sent from and received
  into place with point-to-point sends - no packing kernel, no staging copy on the root).  While chunk
  k is being decoded, chunk k+1 is on the wire and the decoded bytes of chunk k-1 travel back.

With backend "nccl" the transfers are RCCL send/recv over xGMI; the same code runs on gloo/CPU
tensors, which is how tests/test_shard_gloo.py exercises it.  The C-ABI twin of ``decode_stream``
for ONE process driving all GPUs is ``vit_decode_stream_multi`` (csrc/vit_multi.hip).

``decode`` is injected: on a GPU it is the HIP path (decode_batch_dev); the CPU tests inject a
stand-in so that the plumbing can be checked without a device.
"""
import torch
import torch.distributed as dist


def out_bytes(framebits):
    """decoded bytes per frame, as the library writes them (a partial last byte is zero-padded)"""
    return (framebits + 7) // 8


# ---- plain round-robin: frame f -> rank f mod N -------------------------------------------------

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
    """scatter -> decode(local_sym [n, sym_len]) -> [n, (framebits+7)//8] -> gather.  Root returns the
    decoded batch in the original frame order, other ranks None."""
    sym_len = 4 * (framebits + 6)
    local = scatter_frames(sym_all, nframes, sym_len, root, group)
    out_local = decode(local)
    return gather_outputs(out_local, nframes, out_bytes(framebits), root, group)


# ---- chunked pipeline: contiguous blocks, round-robin per chunk ---------------------------------

class StreamPlan:
    """Who decodes which frames.  Chunk k covers frames [k*span, (k+1)*span), span = root_frames +
    (world-1)*chunk_frames; inside it the root owns the first root_frames, then the other ranks (in
    rank order, root skipped) one block of chunk_frames each.  The last chunk may be ragged: blocks
    are filled in that order until the frames run out."""

    def __init__(self, nframes, world, chunk_frames, root_frames=None, root=0):
        if chunk_frames <= 0:
            raise ValueError("chunk_frames must be positive")
        self.nframes, self.world, self.root = int(nframes), int(world), int(root)
        self.chunk_frames = int(chunk_frames)
        self.root_frames = int(chunk_frames if root_frames is None else root_frames)
        if self.root_frames < 0 or (self.world == 1 and self.root_frames == 0):
            raise ValueError("bad root_frames")
        self.span = self.root_frames + (self.world - 1) * self.chunk_frames
        self.nchunks = (self.nframes + self.span - 1) // self.span if self.nframes > 0 else 0

    def _slot(self, rank):
        """offset of `rank`'s block inside a chunk and its nominal length"""
        if rank == self.root:
            return 0, self.root_frames
        pos = rank if rank < self.root else rank - 1  # peers in rank order, root skipped
        return self.root_frames + pos * self.chunk_frames, self.chunk_frames

    def block(self, k, rank):
        """(first frame, count) of the block rank `rank` decodes in chunk k (count may be 0)"""
        off, length = self._slot(rank)
        lo = min(self.nframes, k * self.span + off)
        hi = min(self.nframes, k * self.span + off + length)
        return lo, hi - lo

    def owner(self, f):
        k, o = divmod(int(f), self.span)
        if o < self.root_frames:
            return self.root
        pos = (o - self.root_frames) // self.chunk_frames
        return pos if pos < self.root else pos + 1

    def max_block(self, rank):
        return self._slot(rank)[1]


def _post(ops, group):
    """start a list of (kind, tensor, peer) point-to-point operations; returns their work handles"""
    if not ops:
        return []
    p2p = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer, group) for kind, t, peer in ops]
    return dist.batch_isend_irecv(p2p)


def _wait(works):
    for w in works:
        w.wait()  # nccl: the current stream waits; gloo: the host waits


def decode_stream(sym_all, out_all, nframes, framebits, decode, chunk_frames, root_frames=None, root=0, group=None):
    """Chunked, overlapped scatter -> decode -> gather (see the module docstring).

    sym_all [nframes, 4*(framebits+6)] uint8 and out_all [nframes, (framebits+7)//8] uint8 live on the
    root (other ranks pass None).  `decode(sym_block, out_block)` decodes sym_block's frames into
    out_block (same device; on a GPU it only enqueues on the current stream).  Returns the plan.
    Per chunk and rank at most one send and one receive are in flight in each direction, posted as
    one batch (ncclGroupStart/End under "nccl"): chunk k+1 is received into the other half of a
    double buffer while chunk k is decoded, and the decoded bytes of chunk k leave while k+1 is decoded."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    plan = StreamPlan(nframes, world, chunk_frames, root_frames, root)
    sym_len, olen = 4 * (framebits + 6), out_bytes(framebits)
    is_root = rank == root
    dev = sym_all.device if is_root else _default_device()
    if not is_root:
        mb = plan.max_block(rank)
        rbuf = [torch.empty((mb, sym_len), dtype=torch.uint8, device=dev) for _ in range(2)]
        obuf = [torch.empty((mb, olen), dtype=torch.uint8, device=dev) for _ in range(2)]
    peers = [r for r in range(world) if r != root]

    def post_scatter(k):
        if k >= plan.nchunks:
            return []
        if is_root:
            ops = []
            for r in peers:
                lo, n = plan.block(k, r)
                if n:
                    ops.append(("send", sym_all[lo:lo + n], r))
            return _post(ops, group)
        lo, n = plan.block(k, rank)
        return _post([("recv", rbuf[k & 1][:n], root)] if n else [], group)

    def post_gather(k):
        if is_root:
            ops = []
            for r in peers:
                lo, n = plan.block(k, r)
                if n:
                    ops.append(("recv", out_all[lo:lo + n], r))
            return _post(ops, group)
        lo, n = plan.block(k, rank)
        return _post([("send", obuf[k & 1][:n], root)] if n else [], group)

    sc = post_scatter(0)
    gathers = {}
    for k in range(plan.nchunks):
        _wait(sc)                      # chunk k has arrived (root: its sends of chunk k are done)
        sc = post_scatter(k + 1)       # ordered behind decode k-1 on the current stream: that half is free again
        if k >= 2:
            _wait(gathers.pop(k - 2))  # the output half about to be overwritten has left
        lo, n = plan.block(k, rank)
        if n:
            if is_root:
                decode(sym_all[lo:lo + n], out_all[lo:lo + n])
            else:
                decode(rbuf[k & 1][:n], obuf[k & 1][:n])
        gathers[k] = post_gather(k)    # ordered behind decode k
    for k in sorted(gathers):
        _wait(gathers[k])
    return plan


def _default_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
