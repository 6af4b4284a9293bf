"""Every item of a FULL headline batch (BFV N = 2^15, K = 15: two lanes of 128 ciphertext pairs on two HIP streams, multiply + relinearize) as one digest each.

Test infrastructure for tests/test_gpu_parity.py::test_full_headline_batch_* (round-5 verdict, item 5): the wide strided pass, the grouped XCD order of the
mod-down and the lane interleave exist only at this size.  The inputs are generated on the device from (seed, global row index), so any process -- the product
library in two lanes, the probe build in small single-stream chunks under its fallback switches -- sees the same 256 pairs.

    python tests/full_batch.py lanes            -> JSON list of 256 hex digests on stdout (two lanes of 128, as bench.py runs them)
    python tests/full_batch.py chunks 8         -> the same items, 8 at a time on one stream (small launches: flat workgroup order, narrow strided pass)
"""
import ctypes as C
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BITS = [60] + [58] * 13 + [60]  # bench.py's bfv_n32768_l14
N, TBITS, TOTAL, SEED, KEY_SEED = 32768, 20, 256, 0x5EED, 0xC0FFEE


class Batch:
    def __init__(self):
        import troy_amd as ta
        from troy_amd import capi
        self.ta, self.capi, self.lib = ta, capi, capi.load()
        ta.KernelProvider.initialize(0)
        self.primes = ta.CoeffModulus.Create(N, BITS)
        self.t = ta.PlainModulus.Batching(N, TBITS)
        self.K, self.L = len(self.primes), len(self.primes) - 1
        self.ctx = self.context()
        self.key = ta.DeviceBuffer((self.K - 1) * 2 * self.K * N)
        self.ctx.fill_uniform(self.key, (self.K - 1) * 2 * self.K, self.primes, seed=KEY_SEED)

    def context(self):
        return self.ta.SEALContext(1, N, self.primes, self.t)

    def inputs(self, ctx, first, count):
        """items [first, first + count) of the two operand batches: rows (item, poly, limb) of a, then of b, numbered over the WHOLE batch"""
        L = self.L
        out = []
        for which in (0, 1):
            c = self.ta.Ciphertext(ctx, count, 2, L, False, 1.0, 1, capacity=2)
            ctx.fill_uniform(c.buf, count * 2 * L, self.primes[:L], seed=SEED, row0=which * TOTAL * 2 * L + first * 2 * L)
            out.append(c)
        return out

    def mul_relin(self, ctx, a, b, count, stream=None):
        capi, lib = self.capi, self.lib
        o = self.ta.Ciphertext(ctx, count, 3, self.L, capacity=3)
        sa, sb, so = a.struct(), b.struct(), o.struct()
        capi.check(lib, lib.troyhip_multiply(ctx.h, C.byref(sa), C.byref(sb), C.byref(so), C.c_uint64(count), stream))
        capi.check(lib, lib.troyhip_relinearize(ctx.h, C.byref(so), C.c_void_p(self.key.ptr), C.c_uint64(count), stream))
        return o

    def item(self, ct, index, polys=2):
        words = ct.capacity * self.L * N
        return ct.buf.to_numpy(words, index * words).reshape(ct.capacity, self.L, N)[:polys]

    def digests(self, ct, count):
        return [hashlib.blake2b(self.item(ct, i).tobytes(), digest_size=16).hexdigest() for i in range(count)]

    def lanes(self, keep=False):
        """two lanes of TOTAL / 2 on two streams, one context each, launched back to back (the kernels of the two lanes interleave on the device)"""
        capi, lib, half = self.capi, self.lib, TOTAL // 2
        lanes = []
        for i in range(2):
            cx = self.ctx if i == 0 else self.context()
            h = C.c_void_p()
            capi.check(lib, lib.troyhip_stream_create(C.byref(h)))
            a, b = self.inputs(cx, i * half, half)
            lanes.append((cx, h, a, b))
        self.ta.synchronize()
        outs = [self.mul_relin(cx, a, b, half, h) for cx, h, a, b in lanes]
        for _, h, _, _ in lanes:
            capi.check(lib, lib.troyhip_stream_synchronize(h))
        self.ta.synchronize()
        d = self.digests(outs[0], half) + self.digests(outs[1], half)
        if keep:
            self.kept = (lanes, outs)
        return d

    def chunks(self, size):
        d = []
        for first in range(0, TOTAL, size):
            n = min(size, TOTAL - first)
            a, b = self.inputs(self.ctx, first, n)
            o = self.mul_relin(self.ctx, a, b, n)
            self.ta.synchronize()
            d += self.digests(o, n)
        return d


def oracle_items(batch, indices):
    """the kept lanes' results of `indices` (global item numbers) against the CPU oracle; -> list of mismatching indices"""
    import numpy as np
    from oracle import oracle, ref as R
    O = oracle.Oracle(1, N, batch.primes, batch.t)
    K = batch.K
    O.set_kswitch_key(0, batch.key.to_numpy((K - 1) * 2 * K * N).reshape(K - 1, 2, K, N))
    lanes, outs = batch.kept
    half, bad = TOTAL // 2, []
    for g in indices:
        lane, i = divmod(g, half)
        _, _, a, b = lanes[lane]
        xa, xb = np.ascontiguousarray(batch.item(a, i)), np.ascontiguousarray(batch.item(b, i))
        exp = O.eval(R.OP_RELIN, O.eval(R.OP_MULTIPLY, R.Ct(xa), R.Ct(xb))).data
        if not np.array_equal(batch.item(outs[lane], i), exp):
            bad.append(g)
    return bad


if __name__ == "__main__":
    b = Batch()
    out = b.lanes() if sys.argv[1] == "lanes" else b.chunks(int(sys.argv[2]))
    print(json.dumps(out))
