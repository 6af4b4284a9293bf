"""Batch sharding over the GPUs of one node: one process per GPU, `torch.distributed` with backend "nccl"
(= RCCL over xGMI on ROCm) or "gloo" (CPU rendezvous, tests).  The data path has NO collective: ciphertexts are independent
units.  What travels is (a) once, the keys from rank 0 (`broadcast`), (b) per batch, the scatter of the inputs from rank 0 and
the gather of the outputs back (`scatter_batch` / `gather_batch`) -- and nothing at all when the inputs are produced on the
rank that consumes them (the benchmark).

With NCCL the transfers run on zero-copy torch views of the library's own device allocations (`__cuda_array_interface__`), so
RCCL moves the bytes GPU-to-GPU; with gloo they go through host arrays.  torch is plumbing here: no torch type crosses the C ABI.
"""
import numpy as np

from . import api


def shard_range(total, rank, world):
    """contiguous block of `total` units owned by `rank` (the first total % world ranks get one more)"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class _DevView:
    """exposes a DeviceBuffer range to torch without a copy"""

    def __init__(self, buf, words, offset_words=0):
        self.__cuda_array_interface__ = {"shape": (int(words),), "typestr": "<u8", "data": (buf.ptr + 8 * offset_words, False), "version": 2}


def _dist():
    import torch.distributed as dist
    return dist


def _is_nccl():
    return _dist().get_backend() == "nccl"


def _tensor(buf, words, offset_words=0):
    import torch
    if _is_nccl():
        return torch.as_tensor(_DevView(buf, words, offset_words), device=f"cuda:{torch.cuda.current_device()}").view(torch.int64)
    return torch.from_numpy(buf.to_numpy(words, offset_words).view(np.int64))


def _store(buf, t, words, offset_words=0):
    if _is_nccl():
        return  # the view aliases the buffer
    import ctypes as C
    from . import capi
    a = np.ascontiguousarray(t.numpy().view(np.uint64))
    capi.check(buf.lib, buf.lib.troyhip_copy_h2d(C.c_void_p(buf.ptr + 8 * offset_words), a.ctypes.data_as(C.c_void_p), C.c_size_t(words * 8), None))


def broadcast(buf, src=0):
    """every rank ends up with rank `src`'s content of the DeviceBuffer (keys, once)"""
    api.synchronize()
    t = _tensor(buf, buf.words)
    _dist().broadcast(t, src=src)
    _store(buf, t, buf.words)
    return buf


def scatter_batch(context, full, batch_total, size, limbs, is_ntt_form=False, scale=1.0, correction_factor=1, src=0):
    """`full` (rank `src` only): numpy [batch_total][size][limbs][N].  Returns this rank's shard as a batched Ciphertext."""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(batch_total, rank, world)
    mine = api.Ciphertext(context, max(hi - lo, 1), size, limbs, is_ntt_form, scale, correction_factor, capacity=size)
    per = size * limbs * context.N
    if rank == src:
        full = np.ascontiguousarray(full, dtype=np.uint64).reshape(batch_total, per)
        staged = api.DeviceBuffer.from_numpy(full) if _is_nccl() else None
        reqs = []
        for r in range(world):
            rlo, rhi = shard_range(batch_total, r, world)
            if rhi == rlo:
                continue
            if r == src:
                mine = api.Ciphertext.from_numpy(context, full[rlo:rhi].reshape(rhi - rlo, size, limbs, context.N), is_ntt_form, scale, correction_factor)
                continue
            import torch
            t = _tensor(staged, (rhi - rlo) * per, rlo * per) if staged else torch.from_numpy(full[rlo:rhi].reshape(-1).view(np.int64).copy())
            reqs.append(dist.isend(t, dst=r))
        for q in reqs:
            q.wait()
    elif hi > lo:
        t = _tensor(mine.buf, (hi - lo) * per)
        dist.recv(t, src=src)
        _store(mine.buf, t, (hi - lo) * per)
    api.synchronize()
    return mine


def gather_batch(ct, batch_total, dst=0):
    """inverse of scatter_batch: rank `dst` returns numpy [batch_total][size][limbs][N], the others None"""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(batch_total, rank, world)
    size, limbs, N = ct.size(), ct.limbs, ct.context.N
    per = size * limbs * N
    api.synchronize()
    if rank != dst:
        if hi > lo:
            dense = api.DeviceBuffer.from_numpy(ct.cpu()[: hi - lo])  # dense [shard][size][limbs][N]
            dist.send(_tensor(dense, (hi - lo) * per), dst=dst)
        return None
    out = np.zeros((batch_total, size, limbs, N), dtype=np.uint64)
    for r in range(world):
        rlo, rhi = shard_range(batch_total, r, world)
        if rhi == rlo:
            continue
        if r == dst:
            out[rlo:rhi] = ct.cpu()[: rhi - rlo]
            continue
        tmp = api.DeviceBuffer((rhi - rlo) * per)
        t = _tensor(tmp, (rhi - rlo) * per)
        dist.recv(t, src=r)
        _store(tmp, t, (rhi - rlo) * per)
        out[rlo:rhi] = tmp.to_numpy().reshape(rhi - rlo, size, limbs, N)
    return out
