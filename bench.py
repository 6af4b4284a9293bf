#!/usr/bin/env python3
"""bench.py -- BASELINE metric: ciphertext x ciphertext multiply + relinearize ops/sec, BFV N=2^15 L=14
(K=15 primes), on synthetic uniform ciphertexts resident in HBM, plus the NTT kernel's achieved bandwidth against
the HBM roofline and the CPU path timed beside it.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 5 --warmup 2          (starts the 8 ranks itself: self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path (multiply + relinearize) over one batch of --batch independent ciphertext pairs
per GPU.  Batches shard embarrassingly over ranks (no data-path collective): every rank owns its own
--batch pairs (weak scaling), keys and tables are replicated, only the timing is reduced (max over ranks).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (scheme, N, prime bit sizes, plain-modulus bits)
    "bfv_n32768_l14": (1, 32768, [60] + [58] * 13 + [60], 20),   # BASELINE.json metric config (configs[1] shape at headline size)
    "bfv_n8192_l4": (1, 8192, [40, 36, 36, 36, 40], 20),         # BASELINE.json configs[1]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128, help="ciphertext pairs per GPU per step")
    ap.add_argument("--workload", default="bfv_n32768_l14", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=2, help="split the batch over this many HIP streams (one context each): kernels of different phases overlap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ntt-reps", type=int, default=10)
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

    import numpy as np

    import troy_amd as ta
    from troy_amd import capi

    lib = capi.load()
    ta.KernelProvider.initialize(device)
    scheme, N, bits, tbits = WORKLOADS[args.workload]
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tbits)
    ctx = ta.SEALContext(scheme, N, primes, t)
    K, L = len(primes), len(primes) - 1
    B = args.batch
    nbsk = len(ctx.behz_bases(L)[0])

    # ---- synthetic inputs, generated on the device (fill_uniform_kernel == troy_amd.synth) ----
    # The batch is split over `streams` HIP streams, each with its own context (tables + scratch arena): the kernels of one
    # half run concurrently with the kernels of the other, so VALU-bound phases (BEHZ, fused key-switch pass) overlap
    # memory-bound ones (strided NTT passes).  streams = 1 is the plain single-stream batch.
    S = max(1, min(args.streams, B))
    row0 = rank * B * 4 * L  # every rank owns different ciphertexts
    key = ta.DeviceBuffer((K - 1) * 2 * K * N)
    ctx.fill_uniform(key, (K - 1) * 2 * K, primes, seed=0xC0FFEE)
    lanes = []
    done = 0
    for i in range(S):
        Bi = B // S + (1 if i < B % S else 0)
        cx = ctx if i == 0 else ta.SEALContext(scheme, N, primes, t)
        st = None
        if S > 1:
            h = C.c_void_p()
            capi.check(lib, lib.troyhip_stream_create(C.byref(h)))
            st = h
        ai, bi = ta.Ciphertext(cx, Bi, 2, L), ta.Ciphertext(cx, Bi, 2, L)
        oi = ta.Ciphertext(cx, Bi, 3, L, capacity=3)
        cx.fill_uniform(ai.buf, Bi * 2 * L, primes[:L], seed=0x5EED, row0=row0 + done * 2 * L)
        cx.fill_uniform(bi.buf, Bi * 2 * L, primes[:L], seed=0x5EED, row0=row0 + B * 2 * L + done * 2 * L)
        cx.reserve_scratch(max(cx.scratch_words(0, L, Bi), cx.scratch_words(1, L, Bi)))
        lanes.append((cx, st, Bi, ai.struct(), bi.struct(), oi, ai, bi))
        done += Bi
    ta.synchronize()

    # Odd lanes run half a step out of phase (their relinearize of the previous product is issued while the even lanes
    # multiply): every lane still does exactly one multiply and one relinearize per step.
    pending = {}

    def mul(i):
        cx, st, Bi, sa, sb, oi, _a, _b = lanes[i]
        so = oi.struct()
        capi.check(lib, lib.troyhip_multiply(cx.h, C.byref(sa), C.byref(sb), C.byref(so), C.c_uint64(Bi), st))
        pending[i] = so

    def relin(i):
        cx, st, Bi = lanes[i][:3]
        capi.check(lib, lib.troyhip_relinearize(cx.h, C.byref(pending.pop(i)), C.c_void_p(key.ptr), C.c_uint64(Bi), st))

    if not args.roofline_only:
        for i in range(1, S, 2):
            mul(i)  # prime the out-of-phase lanes (untimed)

    def step():
        for i in range(S):
            (relin if i & 1 else mul)(i)
        for i in range(S):
            (mul if i & 1 else relin)(i)

    def sync_all():
        for _cx, st, *_ in lanes:
            ta.synchronize(st)
        ta.synchronize()

    def barrier():
        sync_all()
        if use_dist:
            _dist().barrier()
        sync_all()

    if args.roofline_only:
        args.steps = args.warmup = 0
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = max(time.perf_counter() - t0, 1e-9)
    barrier()
    own_dt = dt
    if use_dist:
        dt = max_over_ranks(own_dt, backend)
        total_ops = sum_over_ranks(B * args.steps, backend)
        per_rank = gather_floats(B * args.steps / own_dt, backend)
        world_seen = _dist().get_world_size()
    else:
        total_ops = B * args.steps
        per_rank = [B * args.steps / own_dt]
        world_seen = 1
    value = total_ops / dt

    # ---- roofline of the dominant kernel: the batched NTT (both passes of one transform = one "launch" unit) ----
    # shape = the key-switch NTT of this very workload: rows = B * (L+1) * L limb-polynomials, prime index (r / L) % (L+1)
    roofline = None
    if rank == 0:
        rows = B * (L + 1) * L
        D = ta.DeviceBuffer(rows * N)
        out_primes = primes[:L] + [primes[K - 1]]
        ctx.fill_uniform(D, rows, out_primes, seed=1, inner=L)
        pr = np.array(out_primes, dtype=np.uint64)
        timer = C.c_void_p()
        capi.check(lib, lib.troyhip_timer_create(C.byref(timer)))

        def ntt_once(inv):
            capi.check(lib, lib.troyhip_ntt(ctx.h, C.c_void_p(D.ptr), C.c_uint64(rows), pr.ctypes.data_as(C.c_void_p), len(pr), L, inv, None))
        ntt_once(0)
        ntt_once(1)
        ta.synchronize()
        capi.check(lib, lib.troyhip_timer_start(timer, None))
        for i in range(args.ntt_reps):
            ntt_once(i & 1)  # forward / inverse alternate so values stay canonical
        capi.check(lib, lib.troyhip_timer_stop(timer, None))
        ms = C.c_float()
        capi.check(lib, lib.troyhip_timer_elapsed_ms(timer, C.byref(ms)))
        capi.check(lib, lib.troyhip_timer_destroy(timer))
        per_launch_s = ms.value / 1e3 / args.ntt_reps
        algo_bytes = 16.0 * N * rows  # 16 B per coefficient per limb-transform
        achieved = algo_bytes / per_launch_s / 1e9
        # HBM bytes per launch from the committed PMC measurement of this kernel (profiles/ntt_traffic.json:
        # FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 passes); null when the file does not match this workload
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "ntt_traffic.json")))
            if tj.get("N") == N:
                traffic = round(tj["hbm_bytes_per_limb_transform"]["mean"] * rows)
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": "ntt2_kernel (strided pass + contiguous pass = 1 limb-transform)", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": algo_bytes, "launch_us": round(per_launch_s * 1e6, 2),
                    "limb_transforms_per_launch": rows}
        if traffic:  # the two passes' REAL HBM traffic (PMC) over the same live launch time: what rocprof shows the memory system doing
            roofline["hbm_traffic_gbps"] = round(traffic / per_launch_s / 1e9, 1)
            roofline["hbm_traffic_frac"] = round(traffic / per_launch_s / 1e9 / HBM_PEAK_GBPS, 4)
        del D

    # ---- CPU baseline on this box's host cores (rank 0, N=1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(scheme, N, primes, t, L)
        except Exception as e:  # the baseline is informational; never let it kill the bench line
            cpu = {"error": str(e)}

    if rank == 0:
        limb_transforms = 7 * (L + nbsk) + (L + 1) * L + 2 * (L + 1)
        line = {
            "metric": "ct x ct multiply+relinearize ops/sec, BFV N=2^15 L=14; achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3) if args.steps else None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": args.workload, "scheme": "BFV", "N": N, "K": K, "L": L, "Bsk": nbsk, "batch_per_gpu": B,
                       "limb_transforms_per_op": limb_transforms, "parallelism": f"batch-shard x{world}", "streams_per_gpu": S, "rendezvous": backend},
            "ranks": world_seen, "per_rank_ops_per_s": [round(v, 2) for v in per_rank],
            "roofline": roofline, "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.dup2(saved_stdout_fd, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if use_dist:
        _dist().destroy_process_group()


def cpu_baseline(scheme, N, primes, t, L):
    """The reference's own CPU path (oracle/_ref, kind "reference") when the prebuilt library travelled here, else our
    CPU port (kind "port").  Bounded sample: a handful of multiply+relinearize ops on the same synthetic inputs."""
    import numpy as np

    from troy_amd import synth
    xa = synth.uniform_ct(0x5EED, primes[:L], 2, N)[0]
    xb = synth.uniform_ct(0x5EEE, primes[:L], 2, N)[0]
    rk = synth.uniform_kswitch_key(0xC0FFEE, primes, N)
    from oracle import oracle, ref
    reps = 60 if N >= 32768 else 400  # about 12 s of single-core work at either size (0.2 s / 0.03 s per op)
    O = oracle.Oracle(scheme, N, primes, t)
    O.set_kswitch_key(0, rk)
    if ref.available():
        R = ref.Ref(scheme, N, primes, t)
        R.set_kswitch_key(0, rk)
        secs = R.time_mul_relin(ref.Ct(xa), ref.Ct(xb), reps)
        out = {"value": round(reps / secs, 3), "unit": "ops/s", "cores": 1, "kind": "reference",
               "sample": f"{reps} multiply+relinearize ops, 1 thread, reference CPU path (src/troy_cpu.h) built -O2 into oracle/_ref"}
    else:
        secs = O.time_mul_relin(np.ascontiguousarray(xa), np.ascontiguousarray(xb), reps, 1)
        out = {"value": round(reps / secs, 3), "unit": "ops/s", "cores": 1, "kind": "port",
               "sample": f"{reps} multiply+relinearize ops, 1 thread, scalar CPU port (oracle/troy_oracle.cpp, -O3)"}
    # the same work on all host cores, one evaluator per thread over disjoint ciphertexts (the port: its evaluators
    # share nothing but read-only tables), one thread per PHYSICAL core (about 70 MB of working set each at N = 2^15), about 8 s.
    threads = physical_cores()
    if threads > 1:
        reps_all = threads * (30 if N >= 32768 else 200)
        secs = O.time_mul_relin(np.ascontiguousarray(xa), np.ascontiguousarray(xb), reps_all, threads)
        out["all_cores"] = {"value": round(reps_all / secs, 3), "unit": "ops/s", "cores": threads, "kind": "port", "cpu": _cpu_model(),
                            "sample": f"{reps_all} ops over {threads} threads, scalar CPU port (oracle/troy_oracle.cpp, -O3)"}
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
