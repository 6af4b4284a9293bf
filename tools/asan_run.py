import sys, os, time
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
from troy_amd import api, capi
lib = capi.load(os.environ.get('TROY_EMUL_LIB', '/tmp/libtroyhip_emul_asan.so'))  # the sanitizer build made by tools/asan_check.sh
api.KernelProvider.initialize(0, _lib=lib)
import cases
t=time.time()
for K, big in [(2, True), (5, False), (6, True), (9, "small"), (12, True), (15, False), (16, True), (17, True)]:
    cases.check_bfv_multiply_limb_count(K, N=128, batch=1, big=big)
print("limb counts ok", round(time.time()-t,1))
for s in (3, 4, 5, 6, 9, 12):
    cases.check_random_config(s, sizes=(64,128), batch=1)
print("random ok", round(time.time()-t,1))
# single-pass NTT rows (N = 2^15) both directions, a few rows
import numpy as np
from troy_amd import synth
N=32768
kp = api.CoeffModulus.Create(N, [60, 50, 58, 40, 60])
ctx = api.SEALContext(api.BFV, N, kp, api.PlainModulus.Batching(N, 20))
primes = kp[:4]
rows = 16
x = synth.uniform_rows(7, primes, rows, N)
buf = api.DeviceBuffer.from_numpy(x)
ctx.ntt(buf, rows, primes); ctx.ntt(buf, rows, primes, inverse=True)
assert np.array_equal(buf.to_numpy().reshape(rows,N), x)
print("ntt1 ok", round(time.time()-t,1))
