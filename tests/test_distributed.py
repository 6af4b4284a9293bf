"""Batch sharding over ranks: world_size-2 gloo run of bench.py's partition + max-over-ranks logic on
CPU.  No data-path collective exists: ranks own disjoint ciphertext ranges and only the timing is reduced."""
import os
import subprocess
import sys

from conftest import ROOT


def test_shard_ranges_cover_batch():
    sys.path.insert(0, ROOT)
    import bench
    for total in (1, 7, 8, 1024):
        for world in (1, 2, 4, 8):
            spans = [bench.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def test_two_rank_gloo_reduction(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(
        "import sys; sys.path.insert(0, %r)\n"
        "import bench, torch.distributed as dist, os\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "t = bench.max_over_ranks(1.0 + r, backend='gloo')\n"
        "assert abs(t - 2.0) < 1e-9, t\n"
        "tot = bench.sum_over_ranks(3 + r, backend='gloo')\n"
        "assert tot == 7, tot\n"
        "s, e = bench.shard_range(9, r, 2)\n"
        "assert (s, e) == ((0, 5) if r == 0 else (5, 9))\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "os.write(1, ('rank %%d ok\\n' %% r).encode())\n" % ROOT)  # one write per rank: two ranks share the pipe
    import socket
    with socket.socket() as sk:  # a free port: a fixed one can still be in TIME_WAIT from an earlier run
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_two_rank_scatter_compute_gather(tmp_path):
    """troy_amd/dist.py over gloo (world_size 2): rank 0 scatters a batch of 5 ciphertexts (shards of 3 and 2), each rank
    works on its shard (emulator build of the library stands in for the GPU), rank 0 gathers and checks every row; the key
    broadcast is checked the same way.  On a GPU node the same code runs over RCCL (backend "nccl") on device views."""
    import socket
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os; sys.path.insert(0, %r)\n"
        "import numpy as np, torch.distributed as dist\n"
        "from troy_amd import api, capi, dist as tdist, synth\n"
        "lib = capi.load(os.path.join(%r, 'tests', 'emul', 'libtroyhip_emul.so'))\n"
        "api.KernelProvider.initialize(0, _lib=lib)\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "N = 64\n"
        "primes = api.CoeffModulus.Create(N, [40, 40, 40])\n"
        "ctx = api.SEALContext(capi.CKKS, N, primes, 0)\n"
        "full = synth.uniform_ct(9, primes[:2], 2, N, 5) if r == 0 else None\n"
        "mine = tdist.scatter_batch(ctx, full, 5, 2, 2, is_ntt_form=True)\n"
        "assert mine.batch == (3 if r == 0 else 2)\n"
        "api.Evaluator(ctx).negateInplace(mine)\n"
        "out = tdist.gather_batch(mine, 5)\n"
        "dev_full = api.Ciphertext.from_numpy(ctx, full, True) if r == 0 else None\n"   # device-resident batch on rank 0: scattered from views
        "mine2 = tdist.scatter_batch(ctx, dev_full, 5, 2, 2, is_ntt_form=True)\n"
        "assert np.array_equal(mine2.cpu()[:mine.batch], np.where(mine.cpu() == 0, 0, np.array(primes[:2], dtype=np.uint64)[None, None, :, None] - mine.cpu()))\n"
        "back = tdist.gather_batch_device(mine2, 5)\n"
        "assert (back is None) == (r != 0)\n"
        "if r == 0: assert np.array_equal(back.cpu(), full)\n"
        "key = api.DeviceBuffer.from_numpy(np.arange(100, dtype=np.uint64) * (7 if r == 0 else 1))\n"
        "tdist.broadcast(key)\n"
        "assert np.array_equal(key.to_numpy(), np.arange(100, dtype=np.uint64) * 7)\n"
        "if r == 0:\n"
        "    p = np.array(primes[:2], dtype=np.uint64)[None, None, :, None]\n"
        "    assert np.array_equal(out, np.where(full == 0, full, p - full))\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "os.write(1, ('rank %%d ok\\n' %% r).encode())\n" % (ROOT, ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_bench_self_launch_refuses_missing_gpus():
    """`python bench.py --gpus N` starts its own ranks (bench.self_launch); with fewer than N devices visible it must fail
    loudly instead of falling back to a CPU rendezvous (VERDICT r1 #2 / ADVICE r1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has the GPUs; the refusal path needs fewer than 2")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, out.stdout + out.stderr
    assert "--gpus 2 but only" in out.stderr and out.stdout.strip() == ""
