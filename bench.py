#!/usr/bin/env python3
"""bench.py -- BASELINE metric: ciphertext x ciphertext multiply + relinearize ops/sec, BFV N=2^15 L=14
(K=15 primes), on synthetic uniform ciphertexts resident in HBM, plus the NTT kernel's achieved bandwidth against
the HBM roofline and the CPU path timed beside it.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          (starts the 8 ranks itself: self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of --batch independent ciphertexts per GPU.  Batches shard
embarrassingly over ranks (no data-path collective): every rank owns its own --batch units (weak scaling),
keys and tables are replicated, only the timing is reduced (max over ranks).  Rank 0 prints ONE JSON line.

--workload selects the BASELINE.json configuration (default: the metric's own, `bfv_n32768_l14`):
    bfv_n32768_l14         ct x ct multiply + relinearize, BFV N=2^15 K=15            (the headline metric)
    bfv_n8192_l4           the same at BASELINE configs[1] (N=8192, 5 primes)
    ckks_n32768_chain      configs[2]: CKKS N=32768 L=14, multiply -> relinearize -> rescale -> rotate(1), chained to depth 3
    bgv_n65536_relin_rot   configs[3]: BGV N=65536, size-3 ciphertexts, relinearize + rotateRows(1)
    ckks_matmul_128        configs[4]: CKKS 128x128 matmul of app/LinearHelperCKKS.cuh (MatmulHelper), one input row per batch item
Every line carries `roofline` (dominant kernel, algorithmic bytes per SURVEY.md 8d, live HIP-event time), `roofline.per_kernel`
(every kernel of one step, timed by the library's per-launch HIP events) and -- on one rank -- `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ORIG_AFFINITY = None
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BFV, CKKS, BGV = 1, 2, 3

WORKLOADS = {
    # name: scheme, N, prime bit sizes, plain-modulus bits, kind, default batch per GPU, default streams
    "bfv_n32768_l14": dict(scheme=BFV, N=32768, bits=[60] + [58] * 13 + [60], tbits=20, kind="mul_relin", batch=256, streams=2,
                           metric="ct x ct multiply+relinearize ops/sec, BFV N=2^15 L=14; achieved HBM GB/s vs peak"),
    # the headline's shape with primes that fit a double (15 x 49 bits: every transform and the key-switch inner product take the FP64 instances,
    # fpmod.h) -- a secondary line: the 58-bit chain above stays the metric's configuration
    "bfv_n32768_l14_p49": dict(scheme=BFV, N=32768, bits=[49] * 15, tbits=20, kind="mul_relin", batch=256, streams=2,
                               metric="ct x ct multiply+relinearize ops/sec, BFV N=2^15 L=14 with 49-bit primes (FP64-class kernels); achieved HBM GB/s vs peak"),
    "bfv_n8192_l4": dict(scheme=BFV, N=8192, bits=[40, 36, 36, 36, 40], tbits=20, kind="mul_relin", batch=1024, streams=2,
                         metric="ct x ct multiply+relinearize ops/sec, BFV N=8192 L=4 (BASELINE configs[1]); achieved HBM GB/s vs peak"),
    "ckks_n32768_chain": dict(scheme=CKKS, N=32768, bits=[60] + [40] * 13 + [60], tbits=0, kind="ckks_chain", batch=128, streams=1, depth=3,
                              metric="multiply->relinearize->rescale->rotate steps/sec, CKKS N=32768 L=14, chained to depth 3 (BASELINE configs[2])"),
    "bgv_n65536_relin_rot": dict(scheme=BGV, N=65536, bits=[60] + [50] * 13 + [60], tbits=20, kind="relin_rot", batch=128, streams=1,  # BASELINE configs[3]: 1024 ciphertexts over 8 GPUs
                                 metric="relinearize+rotateRows ops/sec on size-3 ciphertexts, BGV N=65536 L=14 (BASELINE configs[3])"),
    # batch 2048 rows: 1.6 GB of input ciphertexts + 1.6 GB of results live (the 256 MiB memory-side cache holds a tenth); 400 steps by default
    # so that the timed region is about half a second
    "ckks_matmul_128": dict(scheme=CKKS, N=8192, bits=[60, 40, 40, 60], tbits=0, kind="matmul", batch=2048, streams=1, dims=(128, 128), steps=400,
                            metric="128x128 HE matmul rows/sec, CKKS N=8192 (app/LinearHelperCKKS.cuh MatmulHelper, BASELINE configs[4])"),
}


def shard_range(total, rank, world):
    """contiguous block partition of `total` independent ciphertexts over `world` ranks"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist


def max_over_ranks(x, backend="nccl"):
    import torch
    dist = _dist()
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(x, backend="nccl"):
    import torch
    dist = _dist()
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def gather_floats(x, backend="nccl"):
    """every rank's value, in rank order (all_gather of one float64)"""
    import torch
    dist = _dist()
    dev = "cuda" if backend == "nccl" else "cpu"
    mine = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: the parent -- before ANY GPU or torch.cuda call, so it never re-execs a
    GPU-initialised process -- starts N ranks as a child `python -m torch.distributed.run`, relays the child's stdout (rank 0's
    ONE JSON line) and returns its exit code.  Fewer than N visible devices is an error, not a silent CPU rendezvous."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()  # counting devices does not initialise the GPU on this image
    if ndev < n and "--allow-gloo" not in sys.argv:
        sys.stderr.write(f"bench.py: --gpus {n} but only {ndev} GPU(s) visible\n")
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def sum_over_ranks(x, backend="nccl"):
    import torch
    dist = _dist()
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([int(x)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


# ---------------------------------------------------------------- workloads: setup -> (step, sync streams, units per step)
class Workload:
    """device-resident synthetic inputs + one `step()` = one pass of the hot path over the rank's batch"""

    def __init__(self, ta, capi, lib, wl, B, streams, rank):
        self.ta, self.capi, self.lib, self.wl, self.B, self.rank = ta, capi, lib, wl, B, rank
        self.scheme, self.N = wl["scheme"], wl["N"]
        self.primes = ta.CoeffModulus.Create(self.N, wl["bits"])
        self.t = ta.PlainModulus.Batching(self.N, wl["tbits"]) if wl["tbits"] else 0
        self.K, self.L = len(self.primes), len(self.primes) - 1
        self.ctx = ta.SEALContext(self.scheme, self.N, self.primes, self.t)
        self.streams = []
        self.units_per_step = B
        getattr(self, "_setup_" + wl["kind"])(max(1, min(streams, B)))
        ta.synchronize()

    def _key(self, ctx, seed):
        K, N = self.K, self.N
        key = self.ta.DeviceBuffer((K - 1) * 2 * K * N)
        ctx.fill_uniform(key, (K - 1) * 2 * K, self.primes, seed=seed)
        return key

    def _ct(self, ctx, batch, size, limbs, seed, row0, ntt, capacity=None, scale=1.0):
        c = self.ta.Ciphertext(ctx, batch, size, limbs, ntt, scale, 1, capacity=capacity or size)
        ctx.fill_uniform(c.buf, batch * size * limbs, self.primes[:limbs], seed=seed, row0=row0)
        return c

    # ---- BFV multiply + relinearize (the headline metric): the batch is split over `S` HIP streams, each with its own context
    # (tables + scratch arena): the kernels of one half run concurrently with the kernels of the other.  Odd lanes run half a
    # step out of phase; every lane still does exactly one multiply and one relinearize per step.
    def _setup_mul_relin(self, S):
        ta, capi, lib, B, L, N = self.ta, self.capi, self.lib, self.B, self.L, self.N
        row0 = self.rank * B * 4 * L  # every rank owns different ciphertexts
        self.key = self._key(self.ctx, 0xC0FFEE)
        lanes, done = [], 0
        for i in range(S):
            Bi = B // S + (1 if i < B % S else 0)
            cx = self.ctx if i == 0 else ta.SEALContext(self.scheme, N, self.primes, self.t)
            st = None
            if S > 1:
                h = C.c_void_p()
                capi.check(lib, lib.troyhip_stream_create(C.byref(h)))
                st = h
            ai = self._ct(cx, Bi, 2, L, 0x5EED, row0 + done * 2 * L, False)
            bi = self._ct(cx, Bi, 2, L, 0x5EED, row0 + B * 2 * L + done * 2 * L, False)
            oi = ta.Ciphertext(cx, Bi, 3, L, capacity=3)
            cx.reserve_scratch(max(cx.scratch_words(0, L, Bi), cx.scratch_words(1, L, Bi)))
            lanes.append((cx, st, Bi, ai.struct(), bi.struct(), oi, ai, bi))
            done += Bi
        self.lanes, self.streams = lanes, [ln[1] for ln in lanes]
        pending = {}

        def mul(i):
            cx, st, Bi, sa, sb, oi = lanes[i][:6]
            so = oi.struct()
            capi.check(lib, lib.troyhip_multiply(cx.h, C.byref(sa), C.byref(sb), C.byref(so), C.c_uint64(Bi), st))
            pending[i] = so

        def relin(i):
            cx, st, Bi = lanes[i][:3]
            capi.check(lib, lib.troyhip_relinearize(cx.h, C.byref(pending.pop(i)), C.c_void_p(self.key.ptr), C.c_uint64(Bi), st))

        def step():
            for i in range(S):
                (relin if i & 1 else mul)(i)
            for i in range(S):
                (mul if i & 1 else relin)(i)

        in_phase = os.environ.get("BENCH_LANE_PHASE") == "same"  # A/B of profiles/r05_overlap.txt: every lane multiplies, then every lane relinearizes

        def step_same():
            for i in range(S):
                mul(i)
            for i in range(S):
                relin(i)

        if in_phase:
            step = step_same

        def prime():
            if in_phase:
                return
            for i in range(1, S, 2):
                mul(i)  # the out-of-phase lanes start with a product to relinearize (untimed)

        def one_lane_step():  # lane 0 alone, in order: what roofline.per_kernel times
            mul(0)
            relin(0)

        def finish_pending():  # the out-of-phase lanes end a step with a product: relinearize it (verification, untimed)
            for i in list(pending):
                relin(i)

        self.finish_pending = finish_pending

        self.step, self.prime, self.profile_step, self.profile_units = step, prime, one_lane_step, lanes[0][2]

    # ---- CKKS multiply -> relinearize -> rescale -> rotate(1), chained `depth` times (levels L .. L-depth+1)
    def _setup_ckks_chain(self, S):
        ta, B, L = self.ta, self.B, self.L
        self.ev = ta.Evaluator(self.ctx)
        scale = float(self.primes[1])  # ~2^40: the product rescales back to ~2^40 at every level
        depth = self.wl["depth"]
        self.x0 = self._ct(self.ctx, B, 2, L, 0x5EED, self.rank * B * 4 * L, True, capacity=3, scale=scale)
        self.bs = [self._ct(self.ctx, B, 2, L - d, 0x7EED + d, self.rank * B * 4 * L, True, scale=scale) for d in range(depth)]
        self.rlk, self.gk = ta.RelinKeys(self.ctx), ta.GaloisKeys(self.ctx)
        self.rlk.keys[0] = self._key(self.ctx, 0xC0FFEE)
        self.gk.keys[ta.GaloisKeys.getIndex(self.ctx.galois_elt_from_step(1))] = self._key(self.ctx, 0xC0FFEF)
        self.units_per_step = B * depth

        def step():
            ev, x = self.ev, self.x0
            for d in range(depth):
                m = ev.multiply(x, self.bs[d])
                ev.relinearizeInplace(m, self.rlk)
                x = ev.rescaleToNext(m)
                x.scale = self.bs[0].scale  # synthetic data: keep the bookkeeping scale at 2^40 exactly (the primes differ from it by < 2^-20)
                ev.rotateVectorInplace(x, 1, self.gk)
            self.last = x

        self.step, self.prime, self.profile_step, self.profile_units = step, (lambda: None), step, B * depth

    # ---- BGV relinearize + rotateRows(1) on size-3 ciphertexts
    def _setup_relin_rot(self, S):
        ta, B, L = self.ta, self.B, self.L
        self.ev = ta.Evaluator(self.ctx)
        self.x3 = self._ct(self.ctx, B, 3, L, 0x5EED, self.rank * B * 3 * L, False, capacity=3)
        self.rlk, self.gk = ta.RelinKeys(self.ctx), ta.GaloisKeys(self.ctx)
        self.rlk.keys[0] = self._key(self.ctx, 0xC0FFEE)
        self.gk.keys[ta.GaloisKeys.getIndex(self.ctx.galois_elt_from_step(1))] = self._key(self.ctx, 0xC0FFEF)

        def step():
            # relinearize(encrypted, keys, destination): every step starts from the same size-3 batch.  The reference's form copies the
            # operand and relinearizes the copy in place; the library reads it where it lies (troyhip_relinearize_to)
            w = self.ev.relinearize(self.x3, self.rlk)
            self.ev.rotateRowsInplace(w, 1, self.gk)
            self.last = w

        self.step, self.prime, self.profile_step, self.profile_units = step, (lambda: None), step, B

    # ---- CKKS 128x128 matmul (MatmulHelper): batch item = one input row; weights encoded once (untimed), inputs synthetic ciphertexts
    def _setup_matmul(self, S):
        import numpy as np
        from troy_amd import app
        B, L, N = self.B, self.L, self.N
        self.ev = self.ta.Evaluator(self.ctx)
        din, dout = self.wl["dims"]
        self.helper = app.MatmulHelper(B, din, dout, N // 2)
        enc = app.CKKSPolyEncoder(self.ctx)
        scale = float(self.primes[1])
        rng = np.random.default_rng(1)
        self.helper.encodeWeights(enc, L, rng.uniform(-1, 1, (din, dout)), scale)
        nblk = (din + self.helper.blockHeight - 1) // self.helper.blockHeight
        self.inputs = [self._ct(self.ctx, B, 2, L, 0x5EED + i, self.rank * B * 2 * L, True, scale=scale) for i in range(nblk)]

        def step():
            self.last = self.helper.matmul(self.ev, self.inputs)

        self.step, self.prime, self.profile_step, self.profile_units = step, (lambda: None), step, B

    def sync_all(self):
        for st in self.streams:
            if st is not None:
                self.ta.synchronize(st)
        self.ta.synchronize()


# ---------------------------------------------------------------- the bench checks its own work
def _item(buf, words, index, shape):
    """host copy of item `index` of a device batch laid out [batch][words]"""
    return buf.to_numpy(words, index * words).reshape(shape)


def _picks(n):
    """first, two interior and last index of a range of n"""
    return sorted({0, n // 3, (2 * n) // 3, n - 1})


def verify(w):
    """four items (first, two interior, last) of the LAST step's results against the CPU oracle on the same inputs (downloaded from the device: the inputs are
    generated there).  Test infrastructure in the checker's role only: nothing here is timed.  -> (True / False, what was compared)"""
    import numpy as np
    from oracle import oracle, ref as R
    kind, N, L, K = w.wl["kind"], w.N, w.L, w.K
    O = oracle.Oracle(w.scheme, N, w.primes, w.t)
    Ct = R.Ct

    def key_host(buf):
        return buf.to_numpy((K - 1) * 2 * K * N).reshape(K - 1, 2, K, N)

    checked = []
    if kind == "mul_relin":
        S = len(w.lanes)
        w.finish_pending()  # odd lanes hold a product waiting for its relinearization: complete their last op (same launch, untimed)
        w.sync_all()
        O.set_kswitch_key(0, key_host(w.key))
        todo = []  # (lane, item): first item of the first lane, last item of the last lane, an interior item of every picked lane -- four checks with two lanes
        for lane in _picks(S):
            Bi = w.lanes[lane][2]
            todo += [(lane, i) for i in sorted({0 if lane == 0 else Bi // 2, Bi // 2, Bi - 1 if lane == S - 1 else Bi // 2})]
        for lane, idx in todo:
            cx, st, Bi, sa, sb, oi, ai, bi = w.lanes[lane]
            xa, xb = _item(ai.buf, ai.bstride, idx, (ai.capacity, L, N))[:2], _item(bi.buf, bi.bstride, idx, (bi.capacity, L, N))[:2]
            got = _item(oi.buf, oi.bstride, idx, (3, L, N))[:2]
            exp = O.eval(R.OP_RELIN, O.eval(R.OP_MULTIPLY, Ct(np.ascontiguousarray(xa)), Ct(np.ascontiguousarray(xb)))).data
            if not np.array_equal(got, exp):
                return False, f"lane {lane} item {idx} differs from the oracle"
            checked.append(f"lane {lane} item {idx}")
        return True, "multiply+relinearize of " + ", ".join(checked) + " (first, interior and last ciphertext pairs of the rank)"
    if kind == "ckks_chain":
        scale, depth = w.bs[0].scale, w.wl["depth"]
        O.set_kswitch_key(0, key_host(w.rlk.keys[0]))
        elt = w.ctx.galois_elt_from_step(1)
        O.set_kswitch_key(elt, key_host(w.gk.keys[w.ta.GaloisKeys.getIndex(elt)]))
        for idx in _picks(w.B):
            y = Ct(np.ascontiguousarray(_item(w.x0.buf, w.x0.bstride, idx, (w.x0.capacity, L, N))[:2]), True, scale)
            for d in range(depth):
                b = Ct(np.ascontiguousarray(_item(w.bs[d].buf, w.bs[d].bstride, idx, (2, L - d, N))), True, scale)
                y = O.eval(R.OP_RESCALE_NEXT, O.eval(R.OP_RELIN, O.eval(R.OP_MULTIPLY, y, b)))
                y.scale = scale
                y = O.eval(R.OP_ROTATE_VECTOR, y, iarg=1)
            got = _item(w.last.buf, w.last.bstride, idx, (w.last.capacity, w.last.limbs, N))[:2]
            if not np.array_equal(got, y.data):
                return False, f"item {idx} differs from the oracle"
            checked.append(str(idx))
        return True, f"depth-{depth} chain of items " + ", ".join(checked)
    if kind == "relin_rot":
        O.set_kswitch_key(0, key_host(w.rlk.keys[0]))
        elt = w.ctx.galois_elt_from_step(1)
        O.set_kswitch_key(elt, key_host(w.gk.keys[w.ta.GaloisKeys.getIndex(elt)]))
        for idx in _picks(w.B):
            x3 = Ct(np.ascontiguousarray(_item(w.x3.buf, w.x3.bstride, idx, (3, L, N))), False)
            exp = O.eval(R.OP_ROTATE_ROWS, O.eval(R.OP_RELIN, x3), iarg=1).data
            got = _item(w.last.buf, w.last.bstride, idx, (w.last.capacity, L, N))[:2]
            if not np.array_equal(got, exp):
                return False, f"item {idx} differs from the oracle"
            checked.append(str(idx))
        return True, "relinearize+rotateRows of items " + ", ".join(checked)
    if kind == "matmul":
        scale = w.inputs[0].scale
        rows, cols = len(w.helper.encodedWeights), len(w.helper.encodedWeights[0])
        for idx in _picks(w.B):
            xs = [Ct(np.ascontiguousarray(_item(a.buf, a.bstride, idx, (2, L, N))), True, scale) for a in w.inputs]
            for j in (0, cols - 1):
                acc = None
                for i in range(rows):
                    pl = w.helper.encodedWeights[i][j].to_numpy(L * N).reshape(L, N)
                    prod = O.eval(R.OP_MULTIPLY_PLAIN_NTT, xs[i], np.ascontiguousarray(pl))
                    acc = prod if acc is None else O.eval(R.OP_ADD, acc, prod)
                out = w.last[j]
                got = _item(out.buf, out.bstride, idx, (out.capacity, L, N))[:2]
                if not np.array_equal(got, acc.data):
                    return False, f"row {idx} output block {j} differs from the oracle"
            checked.append(str(idx))
        return True, "first and last output block of input rows " + ", ".join(checked)
    return None, "no checker for this workload"


# ---------------------------------------------------------------- roofline
def ntt_roofline(ta, capi, lib, w, reps):
    """the batched NTT at the key-switch shape of this workload (rows = B * (L+1) * L limb-polynomials, prime (r / L) % (L+1)),
    forward / inverse alternating, timed with HIP events on the launch stream; algorithmic bytes = 16 B per coefficient per
    limb-transform"""
    import numpy as np
    B, L, K, N, primes = min(w.B, 128), w.L, w.K, w.N, w.primes
    if N > 32768:
        B = min(B, 32)
    rows = B * (L + 1) * L
    D = ta.DeviceBuffer(rows * N)
    out_primes = primes[:L] + [primes[K - 1]]
    w.ctx.fill_uniform(D, rows, out_primes, seed=1, inner=L)
    pr = np.array(out_primes, dtype=np.uint64)
    timer = C.c_void_p()
    capi.check(lib, lib.troyhip_timer_create(C.byref(timer)))

    def ntt_once(inv):
        capi.check(lib, lib.troyhip_ntt(w.ctx.h, C.c_void_p(D.ptr), C.c_uint64(rows), pr.ctypes.data_as(C.c_void_p), len(pr), L, inv, None))

    ntt_once(0)
    ntt_once(1)
    ta.synchronize()
    capi.check(lib, lib.troyhip_ktime_enable(1))
    ntt_once(0)
    ntt_once(1)
    kernels = ktime_report(capi, lib)
    capi.check(lib, lib.troyhip_ktime_enable(0))
    capi.check(lib, lib.troyhip_timer_start(timer, None))
    for i in range(reps):
        ntt_once(i & 1)  # forward / inverse alternate so values stay canonical
    capi.check(lib, lib.troyhip_timer_stop(timer, None))
    ms = C.c_float()
    capi.check(lib, lib.troyhip_timer_elapsed_ms(timer, C.byref(ms)))
    capi.check(lib, lib.troyhip_timer_destroy(timer))
    per_launch_s = ms.value / 1e3 / max(reps, 1)
    algo_bytes = 16.0 * N * rows
    achieved = algo_bytes / per_launch_s / 1e9
    single = any("ntt1_" in k["name"] or "ntt1s_" in k["name"] for k in kernels)
    roof = {"bound": "hbm", "kernel": (("ntt1_fwd_kernel / ntt1_inv_kernel" if N == 32768 else "ntt1s_fwd_kernel / ntt1s_inv_kernel") + " (single pass: one launch = one limb-transform per row)" if single else
                                      "ntt2_kernel (strided pass + contiguous pass = one limb-transform per row)"),
            "kernel_note": "the plain transform, standalone, forward / inverse alternating.  Of this pair only the INVERSE kernel runs inside the timed multiply + relinearize step; the "
                           "step's forward transforms are fused ntt2 passes (tensor, key-switch digit expansion and accumulation): see roofline.in_step for what the step runs",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
            "algorithmic_bytes_per_launch": algo_bytes, "launch_us": round(per_launch_s * 1e6, 2), "limb_transforms_per_launch": rows,
            "launch_batch": B, "launch_batch_cap": "min(batch_per_gpu, 128; 32 above N = 2^15): one lane's key-switch shape, B (L+1) L rows",
            "launch_reps": reps,
            "launch_kernels": [{"name": k["name"], "us": round(k["total_us"] / k["calls"], 1)} for k in kernels],
            "launch_kernels_note": "one forward + one inverse launch under per-kernel events in front of the timed launches (third use of the buffer); launch_us is the "
                                   "mean of the timed back-to-back launches.  A rocprofv3 --stats average over the whole command also contains the first, cold "
                                   "launch of each kernel (about 15 % slower) and that instrumented one"}
    traffic = load_traffic(w.name)
    if traffic and traffic.get("N") == N:
        per_row = traffic.get("hbm_bytes_per_limb_transform", {}).get("ntt1" if single else "ntt2")
        if per_row:
            roof["traffic"] = round(per_row * rows)
            roof["traffic_ratio"] = round(per_row / (16.0 * N), 3)
            roof["traffic_source"] = traffic.get("source")
    del D
    return roof


TRAFFIC_FILES = ("r06_traffic_%s.json", "r05_traffic_%s.json")  # one file per workload (tools/measure_traffic.sh <workload>); the newest whose build id matches


def load_traffic(workload):
    """PMC HBM bytes per kernel (tools/measure_traffic.sh), valid only for the build they were measured on: the file carries the
    library's build id (troyhip_build_id: hash of the sources) and is ignored -- `traffic: null` -- when the loaded library differs"""
    try:
        from troy_amd import capi
        build = capi.build_id()
    except Exception:
        return None
    for name in TRAFFIC_FILES:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name % workload)))
        except Exception:
            continue
        if t.get("build_id") == build:
            return t
    return None


def ktime_report(capi, lib):
    buf = C.create_string_buffer(1 << 16)
    capi.check(lib, lib.troyhip_ktime_report(buf, C.c_size_t(len(buf))))
    out = json.loads(buf.value.decode())
    for k in out:
        n = k["name"].strip()
        if n.startswith("HIP_KERNEL_NAME(") and n.endswith(")"):
            n = n[len("HIP_KERNEL_NAME("):-1]
        k["name"] = n
    return out


def per_kernel(ta, capi, lib, w):
    """every kernel of ONE step of one lane, by the library's per-launch HIP events; algorithmic bytes (compulsory reads + writes
    of the stage) where the table below knows the kernel; HBM traffic from profiles/r06_traffic_<workload>.json (rocprofv3 PMC, tools/measure_traffic.sh)"""
    w.sync_all()
    capi.check(lib, lib.troyhip_ktime_enable(1))
    w.profile_step()
    w.sync_all()
    ks = ktime_report(capi, lib)
    capi.check(lib, lib.troyhip_ktime_enable(0))
    algo = algorithmic_bytes(w, w.profile_units)
    # the forward strided pass has a WIDE form (ntt2.hip n2_wide: 512 threads, twice the columns -- template argument LOGC + 1) that the library takes for
    # launches that fill the chip twice over: same rows, same compulsory bytes, another instance name
    import re
    for n, b in list((algo or {}).items()):
        m = re.match(r"(ntt2(?:_fp)?_kernel<0, 1, )(\d+), (\d+)(, 0, 0, 0>)$", n)
        if m and m.group(2) == "6":
            algo.setdefault("%s%s, %d%s" % (m.group(1), m.group(2), int(m.group(3)) + 1, m.group(4)), b)
    tinfo = load_traffic(w.name) or {}
    traffic = tinfo.get("per_kernel", {})
    out = []
    total_us = sum(k["total_us"] for k in ks) or 1.0
    unknown = [k["name"] for k in ks if algo and k["name"] not in algo and k["total_us"] > 0.02 * total_us]
    if unknown:  # a renamed template instance must not silently drop out of the accounting
        raise RuntimeError("bench.py: algorithmic_bytes() does not know kernels that take more than 2 % of the step: " + ", ".join(unknown))
    for k in ks:
        e = {"name": k["name"], "calls": k["calls"], "us": round(k["total_us"], 1)}
        ab = algo.get(k["name"])
        if ab:
            e["algorithmic_bytes"] = int(ab)
            e["frac"] = round(ab / (k["total_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
        tr = traffic.get(k["name"])
        if tr and ab and w.B == tinfo.get("batch") and w.N == tinfo.get("N"):  # PMC bytes of the same launches at the same batch, scaled to this lane
            units_measured = tinfo["batch"] * (w.wl["depth"] if w.wl["kind"] == "ckks_chain" else 1)  # the PMC pass ran ONE whole step: B units, B x depth for the chain
            e["traffic"] = int(tr["hbm_bytes"] * w.profile_units / units_measured)
            e["traffic_ratio"] = round(e["traffic"] / ab, 3)
        out.append(e)
    return out


def in_step_roofline(kernels):
    """the roofline of what the timed step RUNS (round-4 verdict: the standalone launches of `roofline` time the plain transform at the key-switch shape,
    but the step's forward transforms are fused ntt2 passes -- the tensor pass, the digit-expanding pass, the accumulating pass of the key switch -- and
    only its inverse transforms are the single-pass kernel).  From `per_kernel` (one lane's step under per-launch HIP events): every transform kernel of
    the step with its compulsory stage bytes (operands read once, results written once, the key once per launch), the dominant one, and
    the time-weighted fraction over all of them."""
    ntt = [k for k in kernels if k["name"].startswith(("ntt1", "ntt2")) and k.get("algorithmic_bytes")]
    if not ntt:
        return None
    total_us = sum(k["us"] for k in kernels) or 1.0
    ntt.sort(key=lambda k: -k["us"])
    dom = ntt[0]
    nbytes, nus = sum(k["algorithmic_bytes"] for k in ntt), sum(k["us"] for k in ntt)
    out = {"kernel": dom["name"], "us": dom["us"], "algorithmic_bytes": dom["algorithmic_bytes"], "frac": dom["frac"], "share_of_step": round(dom["us"] / total_us, 3),
           "traffic_ratio": dom.get("traffic_ratio"),
           "transform_kernels": [{"name": k["name"], "us": k["us"], "frac": k["frac"], "share_of_step": round(k["us"] / total_us, 3), "traffic_ratio": k.get("traffic_ratio")} for k in ntt],
           "transform_share_of_step": round(nus / total_us, 3),
           "weighted_frac": round(nbytes / (nus * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
           "note": "fractions of the 8 TB/s HBM peak from the compulsory bytes of each fused stage; the key-switch passes and the tensor pass carry their element-wise work, "
                   "so their bytes per transform exceed 16 B per coefficient"}
    return out


def algorithmic_bytes(w, B):
    """compulsory HBM bytes per kernel NAME over one step of B units (each stage reads its inputs and writes its outputs once;
    the key is read once per launch): the BFV multiply+relinearize path (SURVEY.md 8d's 545 MB per op at cfgNS) here, configs[2] and configs[3] in
    the two functions below."""
    if w.wl["kind"] == "relin_rot":
        return algorithmic_bytes_relin_rot(w, w.B)
    if w.wl["kind"] == "ckks_chain":
        return algorithmic_bytes_ckks_chain(w, w.B)
    if w.wl["kind"] == "matmul":  # the same figures as matmul_roofline: per output block `count` ciphertext columns and plaintexts read, one column written
        P, count = 8.0 * w.N * w.L, len(w.helper.encodedWeights)
        blocks = len(w.helper.encodedWeights[0]) if count else 0
        return {"mul_plain_acc_kernel": blocks * ((2 * count + 2) * w.B + count) * P, "mul_plain_kernel": blocks * count * (4 * w.B + 1) * P, "add_kernel": blocks * max(count - 1, 0) * 6 * w.B * P}
    if w.wl["kind"] != "mul_relin":
        return {}
    N, L = w.N, w.L
    nb = len(w.ctx.behz_bases(L)[0])
    P = 8.0 * N
    logn = N.bit_length() - 1
    t = {}

    def add(name, b):
        t[name] = t.get(name, 0) + b

    two_pass = not 12 <= logn <= 15 or os.environ.get("TROYHIP_NTT") == "twopass"
    k1 = logn - 9 if logn - 9 <= 7 else 7
    logc = 11 - k1
    # multiply: extension of 4 polynomials, forward first passes (q: consumed in place, Bsk), tensor passes, inverse, floor/SK
    fast = "true" if all(int(p) >= 1 << 33 for p in w.ctx.coeff_modulus[:L]) else "false"
    kbx, kb1, kb2 = (L + 4) // 4, (L + 3) // 4, (nb + 3) // 4  # k-blocks: extension (L limbs + r), floor stage 1, stage 2 (|B| limbs + alpha)
    small_x, small_f = kbx <= 2 and nb <= 8, kb2 <= 2           # the everything-in-registers kernels of behz2.hip
    # small bases of narrow primes take the register-resident FP64 form (behz3.hip): every q prime in [2^33, 2^50), every auxiliary prime below 2^50, L <= 6
    fp_behz = _fp_on() and L <= 6 and all((1 << 33) <= int(p) < (1 << FP_MAX_BITS) for p in w.ctx.coeff_modulus[:L]) and all(int(p) < (1 << FP_MAX_BITS) for p in w.ctx.behz_bases(L)[0])
    add(f"behz3_extend_kernel<{L}, {nb}>" if fp_behz else (f"behz2s_extend_kernel<{kbx}, {(nb + 3) // 4}>" if small_x else f"behz2_extend_kernel<{kbx}>"), 2 * 2 * B * (L + nb) * P)
    qs_all = [int(p) for p in w.ctx.coeff_modulus]
    q_primes, special, bsk_primes = qs_all[:L], qs_all[-1:], [int(p) for p in w.ctx.behz_bases(L)[0]]  # the auxiliary base as the library chose it (hostmath.cpp)
    _by_prime_class(add, f"ntt2_kernel<0, 1, {k1}, {logc}, 0, 0, 0>", q_primes + bsk_primes, 2 * (2 * B) * 2 * P)
    _by_prime_class(add, "ntt2_kernel<0, 0, 9, 0, 1, 0, 2>", q_primes + bsk_primes, 7 * B * P)
    md_split = os.environ.get("TROYHIP_MODDOWN", "")[:1] == "s"
    # the library picks the single-pass inverse PER LAUNCH: N = 2^15 and at least four rows per CU (ntt1_supported, ntt1.hip) -- the multiply's three
    # polynomials in both bases are one launch of 3 B (L + |Bsk|) rows, the key switch's two accumulators one of 2 B (L + 1): a small lane batch
    # takes the two-pass kernels for the second while the first still runs single-pass
    env_ntt = os.environ.get("TROYHIP_NTT", "")

    def single(rows):  # ntt1_supported (ntt1.hip): four rows per workgroup slot of the chip -- 1 / 1 / 2 / 4 slots per CU at N = 2^15 .. 2^12
        return not two_pass and (env_ntt == "single" or rows >= 4 * 256 * {15: 1, 14: 1, 13: 2, 12: 4}[logn])
    qs = [int(p) for p in w.ctx.coeff_modulus]

    def n1(tail, primes, per_limb):  # up to three launches by prime class: FP64 rounds for [2^33, 2^50), guard-free integer rounds below 2^58, guarded
        small = f"{logn}, " if logn != 15 else ""  # N = 2^12 .. 2^14: the ntt1s_* instances carry the size
        stem = "ntt1s_inv" if logn != 15 else "ntt1_inv"
        for p in primes:
            if _fp_on() and (1 << 33) <= p < (1 << FP_MAX_BITS):
                add(f"{stem}_fp_kernel<{small}{tail}>", per_limb)
            else:
                add(f"{stem}_kernel<{small}{'true' if (1 << 33) <= p < (1 << 58) else 'false'}, {tail}>", per_limb)
    # multiply: 3 polynomials in both bases
    if single(3 * B * (L + nb)):
        n1("false", q_primes + bsk_primes, 3 * B * 2 * P)
    else:
        _by_prime_class(add, "ntt2_kernel<1, 0, 9, 0, 0, 0, 0>", q_primes + bsk_primes, 3 * B * 2 * P)
        _by_prime_class(add, f"ntt2_kernel<1, 1, {k1}, {logc}, 2, 0, 0>", q_primes + bsk_primes, 3 * B * 2 * P)
    # key switch: 2 accumulators over the L + 1 key primes.  With the mod-down fused (the default) the special limb is transformed on its own and the
    # L data limbs leave through the epilogue: read acc and ct, write ct, plus the special limb once per (ciphertext, accumulator)
    if single(2 * B * (L + 1)):
        fused_md = not md_split and all(p >= 1 << 33 for p in qs[:L])
        n1("false", special, 2 * B * 2 * P)
        if fused_md:
            n1("true", q_primes, 2 * B * 3 * P + 2 * B * P / L)
        else:
            n1("false", q_primes, 2 * B * 2 * P)
    else:
        _by_prime_class(add, "ntt2_kernel<1, 0, 9, 0, 0, 0, 0>", q_primes + special, 2 * B * 2 * P)
        _by_prime_class(add, f"ntt2_kernel<1, 1, {k1}, {logc}, 2, 0, 0>", special, 2 * B * 2 * P)
        if md_split:
            _by_prime_class(add, f"ntt2_kernel<1, 1, {k1}, {logc}, 2, 0, 0>", q_primes, 2 * B * 2 * P)
        else:
            _by_prime_class(add, f"ntt2_kernel<1, 1, {k1}, {logc}, 3, 0, 0>", q_primes, 2 * B * 3 * P + 2 * B * P / L)
    add(f"behz3_floor_sk_kernel<{L}, {nb}>" if fp_behz else f"behz2{'s' if small_f else ''}_floor_sk_kernel<{kb1}, {kb2}, {fast}>", 3 * B * (2 * L + nb) * P)
    # relinearize: digit decomposition + first pass, second pass with the inner product against the key, inverse, mod-down
    _ks_forward_pair(add, w, B, L, logn, False)
    if md_split or (single(2 * B * (L + 1)) and not all(int(p) >= 1 << 33 for p in w.ctx.coeff_modulus[:L])):
        add("ks_moddown_kernel<0>", B * (2 * (L + 1) + 4 * L) * P)
    return t


FP_MAX_BITS = 50  # troy_amd/csrc/fpmod.h: primes below 2^50 run the FP64 instances (ntt2_fp_kernel) unless TROYHIP_FP64=off


def _fp_on():
    return os.environ.get("TROYHIP_FP64", "") != "off"


def _by_prime_class(add, name, primes, per_limb_bytes):
    """bytes of a two-pass kernel split over its two launches: the limbs whose prime lies below 2^50 run the FP64 instance (same template
    arguments, kernel name ntt2_fp_kernel), the others the integer one"""
    nf = sum(1 for p in primes if _fp_on() and int(p) < (1 << FP_MAX_BITS))
    if nf:
        add(name.replace("ntt2_kernel", "ntt2_fp_kernel"), nf * per_limb_bytes)
    if len(primes) - nf:
        add(name, (len(primes) - nf) * per_limb_bytes)


def _ks_forward_pair(add, w, B, l, logn, ckks):
    """the forward pair of ONE key switch of B targets with l limbs: digit-reducing first pass + second pass with the inner product against
    the key, one pair of launches per prime class (FP64 instances for the output primes below 2^50, integer instances for the rest).
    Per class of n output slots: the first pass reads the l digits and writes the expanded rows of its slots; the second reads them, the
    key limbs of its slots (2 components x l digits, once) and -- CKKS -- the NTT-form target row of each data slot, and writes 2 n rows."""
    P = 8.0 * w.N
    k1 = logn - 9 if logn - 9 <= 7 else 7
    logc = 11 - k1
    qs = [int(p) for p in w.ctx.coeff_modulus]
    slots = list(range(l)) + [len(qs) - 1]
    fp_on = os.environ.get("TROYHIP_FP64", "") != "off"
    for cls in (False, True):
        mine = [j for j in slots if (fp_on and qs[j] < (1 << FP_MAX_BITS)) == cls]
        if not mine:
            continue
        exp_rows = sum(l - (1 if ckks and j < l else 0) for j in mine)  # CKKS: the (digit == output prime) row is the NTT-form input itself
        diag = sum(1 for j in mine if ckks and j < l)
        nm = "ntt2_fp_kernel" if cls else "ntt2_kernel"
        add(f"{nm}<0, 1, {k1}, {logc}, 0, {2 if ckks else 1}, 0>", B * l * P + B * exp_rows * P)
        add(f"{nm}<0, 0, 9, 0, 1, 0, {3 if ckks else 1}>", B * exp_rows * P + 2 * len(mine) * l * P + B * diag * P + 2 * B * len(mine) * P)


def _ks_two_pass(add, w, B, L, P, logn, kind):
    """one key switch of B targets with L limbs through the two-pass kernels (BFV kind 3 / BGV kind 4 mod-down epilogue)"""
    k1 = logn - 9 if logn - 9 <= 7 else 7
    logc = 11 - k1
    _ks_forward_pair(add, w, B, L, logn, False)                                                                 # digits read once, (L+1) L expanded limbs written and read, the key once, 2 (L+1) accumulator limbs written
    qs = [int(p) for p in w.ctx.coeff_modulus]
    _by_prime_class(add, "ntt2_kernel<1, 0, 9, 0, 0, 0, 0>", qs[:L] + qs[-1:], 2 * B * 2 * P)                   # first inverse pass of every accumulator limb
    _by_prime_class(add, f"ntt2_kernel<1, 1, {k1}, {logc}, 2, 0, 0>", qs[-1:], 2 * B * 2 * P)                    # the special limb's last pass
    if kind == 4:
        add("ks_bgv_share_kernel", 2 * B * 3 * P)                                                               # special limb read, 128-bit shares written
    _by_prime_class(add, f"ntt2_kernel<1, 1, {k1}, {logc}, {kind}, 0, 0>", qs[:L], 2 * B * 3 * P + 2 * B * (2 if kind == 4 else 1) * P / L)  # acc + ct read, ct written, shares / special limb once


def algorithmic_bytes_relin_rot(w, B):
    """configs[3] (BGV, two-pass sizes): relinearize of a size-3 batch + rotateRows = two key switches, two Galois permutations"""
    if os.environ.get("TROYHIP_MODDOWN", "")[:1] == "s" or os.environ.get("TROYHIP_KS", "")[:1] == "s":
        return {}
    N, L, P = w.N, w.L, 8.0 * w.N
    t = {}

    def add(name, b):
        t[name] = t.get(name, 0) + b

    for _ in range(2):
        _ks_two_pass(add, w, B, L, P, N.bit_length() - 1, 4)
    add("galois_coeff_kernel", 2 * B * L * 2 * P)
    add("copy_strided_kernel", B * L * 2 * P)
    add("zero_strided_kernel", B * L * P)
    return t


def algorithmic_bytes_ckks_chain(w, B):
    """configs[2] at N = 2^15 (single-pass transforms, fused correction): per level l = L .. L - depth + 1 a tensor, a key switch at l, the rescale
    to l - 1, two Galois permutations and a key switch at l - 1"""
    if w.N != 32768 or os.environ.get("TROYHIP_NTT") or os.environ.get("TROYHIP_CORR") or os.environ.get("TROYHIP_KS"):
        return {}
    if 2 * B * (w.L - w.wl.get("depth", 3)) < 4 * 256:  # a launch below four rows per CU takes the two-pass kernels (ntt1_supported): not modelled here
        return {}
    P = 8.0 * w.N
    qs = [int(p) for p in w.ctx.coeff_modulus]
    lean = [(1 << 33) <= p < (1 << 58) for p in qs]
    t = {}

    def add(name, b):
        t[name] = t.get(name, 0) + b

    fpc = [_fp_on() and (1 << 33) <= p < (1 << FP_MAX_BITS) for p in qs]  # the FP64 instances of the single-pass kernels (ntt1_*_fp_kernel)

    def by_class(kernel, tail, slots, per_slot):  # rows of the prime slots split into the FP64, the guard-free and the guarded launch
        nf = sum(1 for i in slots if fpc[i])
        nl = sum(1 for i in slots if lean[i] and not fpc[i])
        if nf:
            add(f"{kernel.replace('_kernel', '_fp_kernel')}<{tail}>", nf * per_slot)
        if nl:
            add(f"{kernel}<true, {tail}>", nl * per_slot)
        if len(slots) - nl - nf:
            add(f"{kernel}<false, {tail}>", (len(slots) - nl - nf) * per_slot)

    def ks(l):
        by_class("ntt1_inv_kernel", "false", range(l), B * 2 * P)                                   # the target to coefficient form, out of place
        # CKKS: the l rows (digit k == output slot) are the NTT-form input itself: not expanded by the first pass, read from the target by the second
        _ks_forward_pair(add, w, B, l, 15, True)
        add("gather_limb_kernel", 2 * B * 2 * P)
        _by_prime_class(add, "ntt2_kernel<1, 0, 9, 0, 0, 0, 0>", qs[-1:], 2 * B * 2 * P)              # the special limb of the accumulators (2 B rows: two-pass)
        _by_prime_class(add, "ntt2_kernel<1, 1, 6, 5, 2, 0, 0>", qs[-1:], 2 * B * 2 * P)
        by_class("ntt1_fwd_kernel", "true", range(l), 2 * B * 3 * P)                                # per row: accumulator + ciphertext read, ciphertext written
        by_class("ntt1_fwd_kernel", "true", range(1), 2 * B * P)                                     # the coefficient-form special limb, once per item (booked on the first slot's class)

    for d in range(w.wl["depth"]):
        l = w.L - d
        add("tensor_kernel<2, 2>", B * l * 7 * P)
        ks(l)
        add("gather_limb_kernel", 2 * B * 2 * P)                                                     # rescale: dropped limb out, inverse, correction transform
        _by_prime_class(add, "ntt2_kernel<1, 0, 9, 0, 0, 0, 0>", qs[l - 1:l], 2 * B * 2 * P)
        _by_prime_class(add, "ntt2_kernel<1, 1, 6, 5, 2, 0, 0>", qs[l - 1:l], 2 * B * 2 * P)
        by_class("ntt1_fwd_kernel", "true", range(l - 1), 2 * B * 2 * P)
        by_class("ntt1_fwd_kernel", "true", range(1), 2 * B * P)
        add("galois_ntt_kernel", 2 * B * (l - 1) * 2 * P)
        add("copy_strided_kernel", B * (l - 1) * 2 * P)
        add("zero_strided_kernel", B * (l - 1) * P)
        ks(l - 1)
    return t


def matmul_roofline(ta, capi, lib, w):
    """configs[4]: the dominant kernel is the fused sum of plaintext products of one output block (mul_plain_acc_kernel: reads `count`
    ciphertext columns and plaintexts, writes one column = (2 count + 2) B + count limb-polynomials), or, with more than 16 input
    blocks, the plain multiply (mul_plain_kernel: (2 + 2) B + 1); timed live by the library's per-launch events over one step"""
    w.sync_all()
    capi.check(lib, lib.troyhip_ktime_enable(1))
    w.step()
    w.sync_all()
    ks = ktime_report(capi, lib)
    capi.check(lib, lib.troyhip_ktime_enable(0))
    k = max(ks, key=lambda e: e["total_us"])
    P = 8.0 * w.N * w.L
    count = len(w.helper.encodedWeights)  # input blocks summed per output block
    per_call = ((2 * count + 2) * w.B + count) * P if "mul_plain_acc" in k["name"] else ((4 * w.B + 1) * P if "mul_plain" in k["name"] else None)
    us = k["total_us"] / k["calls"]
    roof = {"bound": "hbm", "kernel": k["name"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None, "launch_us": round(us, 2), "calls_per_step": k["calls"]}
    if per_call:
        roof.update(achieved=round(per_call / (us * 1e-6) / 1e9, 1), frac=round(per_call / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4), algorithmic_bytes_per_launch=per_call)
    tinfo = load_traffic(w.name)  # PMC bytes of the same kernel over one step at the same batch (tools/measure_traffic.sh), per launch
    tr = (tinfo or {}).get("per_kernel", {}).get(k["name"])
    if tr and tinfo.get("batch") == w.B and tr.get("calls"):
        roof["traffic"] = int(tr["hbm_bytes"] / tr["calls"])
        if per_call:
            roof["traffic_ratio"] = round(roof["traffic"] / per_call, 3)
        roof["traffic_source"] = tinfo.get("source")
    return roof


# ---------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 100; the small ckks_matmul_128 step gets 400 so that the timed region is about 0.5 s)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of items 0 and B-1 of the last step (profiling passes)")
    ap.add_argument("--batch", type=int, default=0, help="units (ciphertext pairs / ciphertexts / input rows) per GPU per step; 0 = the workload's default")
    ap.add_argument("--workload", default="bfv_n32768_l14", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="split the batch over this many HIP streams (one context each); 0 = the workload's default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-kernel", action="store_true")
    ap.add_argument("--ntt-reps", type=int, default=40,
                    help="timed roofline launches (forward / inverse alternating); 40 keeps the two cold launches in front of them below 2 %% of a rocprofv3 average")
    ap.add_argument("--no-roofline", action="store_true", help="skip the roofline launches (PMC passes over the timed step only)")
    ap.add_argument("--roofline-only", action="store_true", help="skip the timed steps: only the roofline NTT launches run (for the rocprofv3 summary of exactly that kernel)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (tests the RCCL path)")
    ap.add_argument("--allow-gloo", action="store_true", help="development only: rendezvous over gloo when fewer GPUs than ranks are visible (never the default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    device = local_rank
    use_dist = world > 1 or args.force_dist
    # RCCL prints a version banner on fd 1; keep stdout clean for the ONE JSON line by routing fd 1 to stderr until then
    sys.stdout.flush()
    saved_stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        import torch
        dist = _dist()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        ndev = torch.cuda.device_count()  # does not initialise the GPU on this image
        device = local_rank % max(ndev, 1)
        if ndev >= world:
            torch.cuda.set_device(device)
            dist.init_process_group("nccl")  # RCCL over xGMI; only used for the barrier and the timing reductions
            max_over_ranks(0.0, "nccl")      # builds the communicator now: a broken RCCL setup fails here, loudly
            backend = "nccl"
        elif args.allow_gloo:                # fewer devices than ranks on a development box: CPU rendezvous, opt-in only
            dist.init_process_group("gloo")
            backend = "gloo"
        else:
            raise SystemExit(f"bench.py: {world} ranks requested but only {ndev} GPU(s) visible (pass --allow-gloo on a development box)")

    import troy_amd as ta
    from troy_amd import capi

    lib = capi.load()
    ta.KernelProvider.initialize(device)
    place = bind_to_device_numa(capi, lib, device)  # one process per GPU: the rank's host threads run on the GPU's NUMA node
    wl = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = wl.get("steps", 100)
    B = args.batch or wl["batch"]
    w = Workload(ta, capi, lib, wl, B, args.streams or wl["streams"], rank)
    w.name = args.workload
    if not args.roofline_only:
        w.prime()

    def barrier():
        w.sync_all()
        if use_dist:
            _dist().barrier()
        w.sync_all()

    if args.roofline_only:
        args.steps = args.warmup = 0
    for _ in range(args.warmup):
        w.step()
    barrier()
    smi = SmiSampler(place.get("pci"))
    smi.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.step()
    w.sync_all()
    dt = max(time.perf_counter() - t0, 1e-9)
    place.update(smi.stop())  # sclk_mhz / power_w averaged over exactly the timed region (rank_devices[])
    barrier()
    placements = [place]
    if use_dist:  # after the timed region: every rank's device, NUMA binding, clock and power
        try:
            placements = [None] * world
            _dist().all_gather_object(placements, place)
        except Exception as e:  # informational: never let the placement report take the run down
            placements = [place, {"note": "all_gather_object failed: " + str(e)[:80]}]
    own_dt = dt
    units = w.units_per_step * args.steps
    if use_dist:
        dt = max_over_ranks(own_dt, backend)
        total_units = sum_over_ranks(units, backend)
        per_rank = gather_floats(units / own_dt, backend)
        world_seen = _dist().get_world_size()
    else:
        total_units = units
        per_rank = [units / own_dt]
        world_seen = 1
    value = total_units / dt

    # the bench checks its own work: items 0 and B-1 of the last step against the CPU oracle (every rank checks its own shard)
    verified, verified_what = None, "skipped (--no-verify / --roofline-only)"
    if not args.no_verify and not args.roofline_only and args.steps > 0:
        try:
            verified, verified_what = verify(w)
        except ImportError as e:  # the checker (oracle/libtroy_oracle.so) did not travel
            verified, verified_what = None, f"oracle unavailable: {e}"
        if use_dist:
            verified = bool(min_over_ranks(1.0 if verified else 0.0, backend) > 0.5) if verified is not None else None

    roofline = None
    if rank == 0 and not args.no_roofline:
        if wl["kind"] == "matmul":
            roofline = matmul_roofline(ta, capi, lib, w)
        else:
            roofline = ntt_roofline(ta, capi, lib, w, args.ntt_reps)
        if not args.no_per_kernel and not args.roofline_only:
            roofline["per_kernel"] = per_kernel(ta, capi, lib, w)
            in_step = in_step_roofline(roofline["per_kernel"])
            if in_step:
                roofline["in_step"] = in_step
                # The ONE fraction a reader sees is that of the kernels the timed step runs (round-5 verdict): time-weighted over the step's transform kernels,
                # algorithmic bytes / HIP-event time of each launch.  The standalone forward / inverse pair (whose forward half never runs in the step) keeps its
                # numbers under roofline.standalone.
                keep = ("kernel", "kernel_note", "achieved", "frac", "traffic", "traffic_ratio", "traffic_source", "algorithmic_bytes_per_launch", "launch_us",
                        "limb_transforms_per_launch", "launch_batch", "launch_batch_cap", "launch_reps", "launch_kernels", "launch_kernels_note")
                roofline["standalone"] = {k: roofline.pop(k) for k in keep if k in roofline}
                ntt = [k for k in roofline["per_kernel"] if k["name"].startswith(("ntt1", "ntt2")) and k.get("algorithmic_bytes")]
                tr = [k.get("traffic") for k in ntt]
                roofline.update(
                    kernel="the transform kernels of the timed step, time-weighted (%d kernels, %.0f %% of the step; dominant: %s at %.3f)"
                           % (len(ntt), 100 * in_step["transform_share_of_step"], in_step["kernel"], in_step["frac"]),
                    achieved=round(in_step["weighted_frac"] * HBM_PEAK_GBPS, 1), frac=in_step["weighted_frac"],
                    traffic=int(sum(tr)) if tr and all(t is not None for t in tr) else None,
                    algorithmic_bytes_per_launch=int(sum(k["algorithmic_bytes"] for k in ntt)), launch_us=round(sum(k["us"] for k in ntt), 1),
                    launch_note="one lane's step (%d units) under per-launch HIP events on the launch stream: sum over the transform kernels" % w.profile_units)
            # the metric's second half, "achieved HBM GB/s vs peak", for the OPERATION: HBM bytes of one whole step (PMC counters of this build,
            # tools/measure_traffic.sh; the compulsory bytes beside them) over the step time of the timed region
            pk = roofline["per_kernel"]
            if args.steps and pk:
                scale = w.units_per_step / float(w.profile_units)
                algo_step = sum(k.get("algorithmic_bytes", 0) for k in pk) * scale
                pmc = [k.get("traffic") for k in pk if k.get("algorithmic_bytes")]
                step_s = dt / args.steps
                st = {"algorithmic_bytes_per_step": int(algo_step), "algorithmic_GBps": round(algo_step / step_s / 1e9, 1),
                      "bytes_per_step": None, "achieved_GBps": None, "frac": None, "peak": HBM_PEAK_GBPS, "units_per_step": w.units_per_step}
                if pmc and all(t is not None for t in pmc):
                    b = sum(pmc) * scale
                    st.update(bytes_per_step=int(b), achieved_GBps=round(b / step_s / 1e9, 1), frac=round(b / step_s / 1e9 / HBM_PEAK_GBPS, 4),
                              bytes_per_unit=int(b / w.units_per_step), source=(load_traffic(w.name) or {}).get("source"))
                else:
                    st["note"] = "no PMC traffic file for this build (profiles/%s): achieved_GBps needs tools/measure_traffic.sh on the same build" % (TRAFFIC_FILES[0] % w.name)
                roofline["step"] = st

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(w)
        except Exception as e:  # the baseline is informational; never let it kill the bench line
            cpu = {"error": str(e)}

    if rank == 0:
        L, K, N = w.L, w.K, w.N
        cfg = {"workload": args.workload, "scheme": {BFV: "BFV", CKKS: "CKKS", BGV: "BGV"}[w.scheme], "N": N, "K": K, "L": L, "batch_per_gpu": B,
               "parallelism": f"batch-shard x{world}", "streams_per_gpu": len(w.streams) or 1, "rendezvous": backend}
        if wl["kind"] == "mul_relin":
            nbsk = len(w.ctx.behz_bases(L)[0])
            cfg.update(Bsk=nbsk, limb_transforms_per_op=7 * (L + nbsk) + (L + 1) * L + 2 * (L + 1))
        line = {
            "metric": wl["metric"], "value": round(value, 2), "unit": {"mul_relin": "ops/s", "ckks_chain": "steps/s", "relin_rot": "ops/s", "matmul": "rows/s"}[wl["kind"]],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3) if args.steps else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic", "config": cfg, "ranks": world_seen, "per_rank_ops_per_s": [round(v, 2) for v in per_rank],
            "verified": verified, "verified_what": verified_what, "build_id": capi.build_id(), "rank_devices": placements,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if use_dist:
        _dist().destroy_process_group()
    if verified is False:
        sys.stderr.write("bench.py: VERIFICATION FAILED: " + verified_what + "\n")
        sys.exit(1)


def cpu_baseline(w):
    """The reference's own CPU path (oracle/_ref, kind "reference") when the prebuilt library travelled here, else our CPU port
    (kind "port"), on a BOUNDED sample of the same workload (about 10-20 s), 1 thread; plus -- for the multiply+relinearize
    workloads -- the port on all physical cores."""
    import numpy as np
    from oracle import oracle, ref
    from troy_amd import synth
    if ORIG_AFFINITY:
        os.sched_setaffinity(0, ORIG_AFFINITY)  # the CPU baseline is about the whole box, not the GPU's NUMA node
    scheme, N, primes, t, L, kind = w.scheme, w.N, w.primes, w.t, w.L, w.wl["kind"]
    use_ref = ref.available()
    E = (ref.Ref if use_ref else oracle.Oracle)(scheme, N, primes, t)
    what = "reference CPU path (src/troy_cpu.h) built -O2 into oracle/_ref" if use_ref else "scalar CPU port (oracle/troy_oracle.cpp, -O3)"
    knd = "reference" if use_ref else "port"
    rk = synth.uniform_kswitch_key(0xC0FFEE, primes, N)
    E.set_kswitch_key(0, rk)
    R, Ct = ref, ref.Ct
    if kind == "mul_relin":
        xa = synth.uniform_ct(0x5EED, primes[:L], 2, N)[0]
        xb = synth.uniform_ct(0x5EEE, primes[:L], 2, N)[0]
        reps = 8 if N >= 32768 else 60  # about 2 s of single-core work at either size (0.2 s / 0.03 s per op)
        if use_ref:
            secs = E.time_mul_relin(Ct(xa), Ct(xb), reps)
        else:
            secs = E.time_mul_relin(np.ascontiguousarray(xa), np.ascontiguousarray(xb), reps, 1)
        out = {"value": round(reps / secs, 3), "unit": "ops/s", "cores": 1, "kind": knd, "sample": f"{reps} multiply+relinearize ops, 1 thread, {what}"}
        # the same work on all host cores.  With the reference built here (oracle/_ref): N single-threaded PROCESSES, one reference
        # evaluator each, pinned to distinct physical cores spread over the sockets, started together -- no shared allocator, no shared pages
        # (round 3's threads of the port fell from 75 to 51 ops/s between 32 and 128 threads: allocator and page-fault contention in one
        # address space, not the algorithm).  Without the reference: the port's thread pool as before.
        cores = physical_cores()
        if cores > 1 and use_ref:
            try:
                out["all_cores"] = _all_cores_reference(scheme, N, primes, t, L, cores)
            except Exception as e:  # informational
                out["all_cores"] = {"error": str(e)[:200]}
        elif cores > 1:
            O = oracle.Oracle(scheme, N, primes, t)
            O.set_kswitch_key(0, rk)
            sweep = {}
            for threads in sorted({min(cores, n) for n in (32, 64, cores)}):
                reps_all = threads * (3 if N >= 32768 else 30)
                secs = O.time_mul_relin(np.ascontiguousarray(xa), np.ascontiguousarray(xb), reps_all, threads)
                sweep[threads] = round(reps_all / secs, 3)
            best = max(sweep, key=sweep.get)
            out["all_cores"] = {"value": sweep[best], "unit": "ops/s", "cores": best, "kind": "port", "cpu": _cpu_model(), "physical_cores": cores,
                                "thread_sweep": {str(k): v for k, v in sweep.items()},
                                "sample": f"best of the thread counts {sorted(sweep)}: {best} threads, one evaluator per thread over disjoint ciphertexts, scalar CPU port (oracle/troy_oracle.cpp, -O3)"}
        return out
    if kind == "ckks_chain":
        scale = float(primes[1])
        E.set_kswitch_key(E.elt_from_step(1), synth.uniform_kswitch_key(0xC0FFEF, primes, N))
        depth = w.wl["depth"]
        x = Ct(synth.uniform_ct(0x5EED, primes[:L], 2, N)[0], True, scale)
        bs = [Ct(synth.uniform_ct(0x7EED + d, primes[:L - d], 2, N)[0], True, scale) for d in range(depth)]
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 8.0:
            y = x
            for d in range(depth):
                y = E.eval(R.OP_RESCALE_NEXT, E.eval(R.OP_RELIN, E.eval(R.OP_MULTIPLY, y, bs[d])))
                y.scale = scale
                y = E.eval(R.OP_ROTATE_VECTOR, y, iarg=1)
                n += 1
        secs = time.perf_counter() - t0
        return {"value": round(n / secs, 3), "unit": "steps/s", "cores": 1, "kind": knd,
                "sample": f"{n} multiply->relinearize->rescale->rotate steps ({n // depth} chains of depth {depth}), 1 thread, {what}"}
    if kind == "relin_rot":
        E.set_kswitch_key(E.elt_from_step(1), synth.uniform_kswitch_key(0xC0FFEF, primes, N))
        x3 = Ct(synth.uniform_ct(0x5EED, primes[:L], 3, N)[0], False)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 8.0:
            E.eval(R.OP_ROTATE_ROWS, E.eval(R.OP_RELIN, x3), iarg=1)
            n += 1
        secs = time.perf_counter() - t0
        return {"value": round(n / secs, 3), "unit": "ops/s", "cores": 1, "kind": knd, "sample": f"{n} relinearize+rotateRows ops on one size-3 ciphertext, 1 thread, {what}"}
    if kind == "matmul":
        # one input row = ceil(in/h) x ceil(out/w) multiplyPlain (NTT form) + the additions, as MatmulHelper::matmul does it
        scale = float(primes[1])
        h = w.helper
        rows_blk, cols_blk = len(h.encodedWeights), len(h.encodedWeights[0])
        xs = [Ct(synth.uniform_ct(0x5EED + i, primes[:L], 2, N)[0], True, scale) for i in range(rows_blk)]
        pl = synth.uniform_rows(0x9999, primes[:L], L, N)
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 8.0:
            for j in range(cols_blk):
                acc = None
                for i in range(rows_blk):
                    p = E.eval(R.OP_MULTIPLY_PLAIN_NTT, xs[i], pl)
                    acc = p if acc is None else E.eval(R.OP_ADD, acc, p)
            n += 1
        secs = time.perf_counter() - t0
        return {"value": round(n / secs, 3), "unit": "rows/s", "cores": 1, "kind": knd, "sample": f"{n} input rows ({rows_blk}x{cols_blk} multiplyPlain + adds each), 1 thread, {what}"}
    return None


_WORKER = r"""
import os, sys, time
cpu, root = int(sys.argv[1]), sys.argv[2]
try:
    os.sched_setaffinity(0, {cpu})
except OSError:
    pass
sys.path.insert(0, root)
import numpy as np
from oracle import ref
from troy_amd import synth
scheme, N, t, L = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
primes = [int(x) for x in sys.argv[7].split(",")]
E = ref.Ref(scheme, N, primes, t)
E.set_kswitch_key(0, np.load(sys.argv[8], mmap_mode="r"))            # the relinearization key, generated once by the parent
a = ref.Ct(synth.uniform_ct(0x5EED + cpu, primes[:L], 2, N)[0])   # disjoint ciphertexts per worker
b = ref.Ct(synth.uniform_ct(0x6EED + cpu, primes[:L], 2, N)[0])
E.time_mul_relin(a, b, 1)                                          # first touch of every table and scratch page
print("ready", flush=True)
for line in sys.stdin:
    w = line.split()
    if not w or w[0] == "quit":
        break
    t0, reps = float(w[1]), int(w[2])
    while time.time() < t0:
        pass
    start = time.time()
    E.time_mul_relin(a, b, reps)
    print("done %.6f %.6f" % (start, time.time()), flush=True)
"""


def _all_cores_reference(scheme, N, primes, t, L, cores):
    """multiply + relinearize on the reference's own CPU path (oracle/_ref) in one single-threaded process per physical core: the processes set
    up once, then run timed rounds with 32, 64 and all of them active (started together on a wall-clock mark); rate = ops of the round / (last
    end - first start).  About 10 s in all."""
    import subprocess
    allowed = sorted(os.sched_getaffinity(0))
    seen, cpus = set(), []
    for cpu in allowed:  # one logical CPU per physical core, in CPU order (sockets / CCDs in their natural order)
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read().strip()
            core = min(int(x) for part in sib.split(",") for x in part.split("-"))
        except (OSError, ValueError):
            core = cpu
        if core not in seen:
            seen.add(core)
            cpus.append(cpu)
    n = len(cpus)
    import numpy as np
    from troy_amd import synth
    keyfile = f"/dev/shm/troy_bench_key_{os.getpid()}.npy" if os.path.isdir("/dev/shm") else os.path.join(ROOT, f".troy_bench_key_{os.getpid()}.npy")
    np.save(keyfile, synth.uniform_kswitch_key(0xC0FFEE, primes, N))
    args = [str(scheme), str(N), str(t), str(L), ",".join(str(p) for p in primes), keyfile]
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, str(c), ROOT] + args, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for c in cpus]
    try:
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("a baseline worker did not start")
        reps = 2 if N >= 32768 else 16
        sweep = {}
        for count in sorted({min(n, c) for c in (32, 64, n)}):
            step = n / count
            active = [procs[int(i * step)] for i in range(count)]  # spread evenly over the core list: both sockets at every count
            t0 = time.time() + 0.2
            for p in active:
                p.stdin.write(f"run {t0:.6f} {reps}\n")
                p.stdin.flush()
            spans = [p.stdout.readline().split() for p in active]
            first, last = min(float(x[1]) for x in spans), max(float(x[2]) for x in spans)
            sweep[count] = round(count * reps / (last - first), 3)
    finally:
        for p in procs:
            try:
                p.stdin.write("quit\n")
                p.stdin.flush()
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()
        try:
            os.remove(keyfile)
        except OSError:
            pass
    best = max(sweep, key=sweep.get)
    return {"value": sweep[best], "unit": "ops/s", "cores": best, "kind": "reference", "cpu": _cpu_model(), "physical_cores": n,
            "process_sweep": {str(k): v for k, v in sweep.items()},
            "sample": f"best of {sorted(sweep)} single-threaded processes ({best}), each the reference CPU path (src/troy_cpu.h, oracle/_ref) on its own physical core "
                      f"over its own ciphertexts, {reps} multiply+relinearize ops per process and round, started together"}


def bind_to_device_numa(capi, lib, device):
    """PCI address and NUMA node of the rank's GPU; the process is bound to that node's CPUs (launches, the MAX-over-ranks timing and the
    CPU baseline then do not cross sockets).  Best effort: a box without the sysfs entries just reports what it found."""
    global ORIG_AFFINITY
    ORIG_AFFINITY = os.sched_getaffinity(0)
    info = {"device": device, "pci": None, "numa_node": None, "cpus_bound": None}
    try:
        buf = C.create_string_buffer(32)
        capi.check(lib, lib.troyhip_device_pci_bus_id(device, buf, C.c_size_t(len(buf))))
        info["pci"] = buf.value.decode().lower()
        node = int(open(f"/sys/bus/pci/devices/{info['pci']}/numa_node").read().strip())
        info["numa_node"] = node
        if node >= 0 and os.environ.get("TROYHIP_BENCH_NO_BIND") is None:
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
                info["cpus_bound"] = len(cpus)
    except Exception as e:  # noqa: BLE001 -- informational only
        info["note"] = str(e)[:80]
    return info


class SmiSampler:
    """Clock and power of the rank's GPU over the timed region, from a side thread that only READS the amdgpu sysfs files of the device's PCI
    function (hwmon freq1_input = shader clock in Hz, power1_average / power1_input in microwatts; pp_dpm_sclk as the fallback) -- no GPU call,
    no child process, nothing that could perturb the step (a file read every 20 ms on a host thread).  Best effort: a box without the files
    reports `{"note": ...}` instead of numbers."""

    def __init__(self, pci, period_s=0.02, sysfs="/sys/bus/pci/devices"):
        import glob
        self.period, self.sclk, self.power, self.note = period_s, [], [], None
        self._stop = threading.Event()
        self._thread = None
        base = f"{sysfs}/{pci}" if pci else None
        hw = sorted(glob.glob(base + "/hwmon/hwmon*")) if base else []
        self.f_sclk = next((h + "/freq1_input" for h in hw if os.path.exists(h + "/freq1_input")), None)
        self.f_power = next((h + "/" + n for h in hw for n in ("power1_average", "power1_input") if os.path.exists(h + "/" + n)), None)
        self.f_dpm = base + "/pp_dpm_sclk" if base and os.path.exists(base + "/pp_dpm_sclk") else None
        if not (self.f_sclk or self.f_dpm or self.f_power):
            self.note = "no amdgpu hwmon / pp_dpm_sclk files under " + str(base)

    def _read_once(self):
        try:
            if self.f_sclk:
                self.sclk.append(int(open(self.f_sclk).read().strip()) / 1e6)
            elif self.f_dpm:
                for ln in open(self.f_dpm).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        self.sclk.append(float(ln.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip()))
            if self.f_power:
                self.power.append(int(open(self.f_power).read().strip()) / 1e6)
        except (OSError, ValueError, IndexError):
            pass

    def _run(self):
        while not self._stop.wait(self.period):
            self._read_once()

    def start(self):
        if self.note is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        if self.note is None and not self.sclk and not self.power:
            self._read_once()  # a timed region shorter than one period still reports a reading (taken right after it: the clock is already on its way down)
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        out = {"samples": max(len(self.sclk), len(self.power)), "period_ms": round(self.period * 1e3, 1), "source": "amdgpu sysfs (hwmon freq1_input / power1_average), read-only side thread over the timed region"}
        if self.sclk:
            out.update(sclk_mhz=round(sum(self.sclk) / len(self.sclk), 1), sclk_mhz_min=round(min(self.sclk), 1), sclk_mhz_max=round(max(self.sclk), 1))
        if self.power:
            out.update(power_w=round(sum(self.power) / len(self.power), 1), power_w_max=round(max(self.power), 1))
        if self.note:
            out["note"] = self.note
        return out


def physical_cores():
    """physical cores this process may run on (SMT siblings counted once): SURVEY.md 8(d) asks for all of them"""
    allowed = os.sched_getaffinity(0)
    cores = set()
    for cpu in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read().strip()
            cores.add(min(int(x) for part in sib.split(",") for x in part.split("-")))
        except (OSError, ValueError):
            cores.add(cpu)
    return max(1, len(cores))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
