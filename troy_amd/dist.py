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


def _shard_tensor(ct, count):
    """torch view of the first `count` ciphertexts of a batched Ciphertext as ONE contiguous int64 tensor (RCCL path): the device
    buffer itself when the batch is dense (capacity == size), else a device-side compaction (torch slice on the strided view)"""
    per_cap = ct.capacity * ct.limbs * ct.context.N
    per = ct.size() * ct.limbs * ct.context.N
    t = _tensor(ct.buf, count * per_cap)
    if per_cap == per:
        return t
    return t.view(count, per_cap)[:, :per].contiguous().view(-1)


def scatter_batch(context, full, batch_total, size, limbs, is_ntt_form=False, scale=1.0, correction_factor=1, src=0):
    """Rank `src` holds the whole batch -- a numpy array [batch_total][size][limbs][N] or a device-resident dense `api.Ciphertext` --
    and every rank gets its contiguous shard as a batched Ciphertext.  RCCL: isend straight from device views (a host array is
    staged shard by shard, never as a whole); gloo: through host arrays."""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(batch_total, rank, world)
    per = size * limbs * context.N
    if rank == src:
        on_device = isinstance(full, api.Ciphertext)
        if on_device:
            if full.capacity != full.size() or full.size() != size or full.limbs != limbs:
                raise ValueError("scatter_batch: the device batch must be dense [batch][size][limbs][N]")
        else:
            full = np.ascontiguousarray(full, dtype=np.uint64).reshape(batch_total, per)
        reqs, keep = [], []
        mine = None
        for r in range(world):
            rlo, rhi = shard_range(batch_total, r, world)
            if rhi == rlo:
                continue
            if r == src:
                if on_device:
                    mine = api.Ciphertext(context, rhi - rlo, size, limbs, is_ntt_form, scale, correction_factor, capacity=size)
                    mine.buf.copy_from(full.buf, (rhi - rlo) * per, src_offset_words=rlo * per)
                else:
                    mine = api.Ciphertext.from_numpy(context, full[rlo:rhi].reshape(rhi - rlo, size, limbs, context.N), is_ntt_form, scale, correction_factor)
                continue
            if _is_nccl():
                if on_device:
                    t = _tensor(full.buf, (rhi - rlo) * per, rlo * per)
                else:
                    staged = api.DeviceBuffer.from_numpy(full[rlo:rhi])  # shard-sized staging, alive until the send completed
                    keep.append(staged)
                    t = _tensor(staged, (rhi - rlo) * per)
            else:
                import torch
                host = full.cpu().reshape(-1)[rlo * per:rhi * per] if on_device else full[rlo:rhi].reshape(-1)
                t = torch.from_numpy(np.ascontiguousarray(host).view(np.int64).copy())
            reqs.append(dist.isend(t, dst=r))
        for q in reqs:
            q.wait()
        if mine is None:
            mine = api.Ciphertext(context, 1, size, limbs, is_ntt_form, scale, correction_factor, capacity=size)
    else:
        mine = api.Ciphertext(context, max(hi - lo, 1), size, limbs, is_ntt_form, scale, correction_factor, capacity=size)
        if hi > lo:
            t = _tensor(mine.buf, (hi - lo) * per)
            dist.recv(t, src=src)
            _store(mine.buf, t, (hi - lo) * per)
    api.synchronize()
    return mine


def gather_batch_device(ct, batch_total, dst=0):
    """inverse of scatter_batch, result left on the device: rank `dst` returns a dense batched Ciphertext of all `batch_total` items
    (received straight into its buffer over RCCL), the others None"""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(batch_total, rank, world)
    size, limbs, N = ct.size(), ct.limbs, ct.context.N
    per = size * limbs * N
    api.synchronize()
    if rank != dst:
        if hi > lo:
            if _is_nccl():
                dist.send(_shard_tensor(ct, hi - lo), dst=dst)  # device view (or device-side compaction): no host bounce
            else:
                import torch
                dist.send(torch.from_numpy(np.ascontiguousarray(ct.cpu()[: hi - lo]).reshape(-1).view(np.int64).copy()), dst=dst)
        return None
    out = api.Ciphertext(ct.context, batch_total, size, limbs, ct.is_ntt_form, ct.scale, ct.correction_factor, capacity=size)
    for r in range(world):
        rlo, rhi = shard_range(batch_total, r, world)
        if rhi == rlo:
            continue
        if r == dst:
            if ct.capacity == size:
                out.buf.copy_from(ct.buf, (rhi - rlo) * per, dst_offset_words=rlo * per)
            else:
                part = api.DeviceBuffer.from_numpy(ct.cpu()[: rhi - rlo])
                out.buf.copy_from(part, (rhi - rlo) * per, dst_offset_words=rlo * per)
            continue
        t = _tensor(out.buf, (rhi - rlo) * per, rlo * per)
        dist.recv(t, src=r)
        _store(out.buf, t, (rhi - rlo) * per, rlo * per)
    api.synchronize()
    return out


def gather_batch(ct, batch_total, dst=0):
    """as gather_batch_device, downloaded: rank `dst` returns numpy [batch_total][size][limbs][N], the others None"""
    out = gather_batch_device(ct, batch_total, dst)
    return None if out is None else out.cpu()
