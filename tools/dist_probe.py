"""single-rank NCCL (RCCL) check of troy_amd/dist.py on a GPU box: zero-copy torch view of a DeviceBuffer + broadcast"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import troy_amd as ta
from troy_amd import api, capi, dist as tdist, synth
torch.cuda.set_device(0)
ta.KernelProvider.initialize(0)
dist.init_process_group("nccl")
buf = api.DeviceBuffer.from_numpy(np.arange(1000, dtype=np.uint64))
t = tdist._tensor(buf, 1000)
assert t.is_cuda and int(t[999]) == 999 and t.data_ptr() == buf.ptr
t[5] = 12345  # the view aliases the library's allocation
assert int(buf.to_numpy()[5]) == 12345
tdist.broadcast(buf)
N = 4096
primes = api.CoeffModulus.Create(N, [40, 40, 40])
ctx = api.SEALContext(capi.CKKS, N, primes, 0)
full = synth.uniform_ct(9, primes[:2], 2, N, 4)
mine = tdist.scatter_batch(ctx, full, 4, 2, 2, is_ntt_form=True)
api.Evaluator(ctx).negateInplace(mine)
out = tdist.gather_batch(mine, 4)
p = np.array(primes[:2], dtype=np.uint64)[None, None, :, None]
assert np.array_equal(out, np.where(full == 0, full, p - full))
dist.destroy_process_group()
print("nccl single-rank dist probe ok")
