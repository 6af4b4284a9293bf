"""Per-kernel times of one BFV multiply batch (BASELINE headline shape N=2^15, L=14) under the BEHZ kernel forms
(TROYHIP_BEHZ = default 8-shift rows | mfma1 | valu), from the library's own per-launch HIP-event timing (troyhip_ktime_*).
usage: python tools/behz_cmp.py [batch]      (development tool)"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(B):
    import troy_amd as ta
    from troy_amd import api, capi
    api.KernelProvider.initialize(0)
    lib = api.KernelProvider.lib() if hasattr(api.KernelProvider, "lib") else capi.load()
    N, bits = 32768, [60] + [58] * 13 + [60]
    primes = ta.CoeffModulus.Create(N, bits)
    ctx = ta.SEALContext(capi.BFV, N, primes, ta.PlainModulus.Batching(N, 20))
    L = len(primes) - 1
    a, b = api.Ciphertext(ctx, B, 2, L), api.Ciphertext(ctx, B, 2, L)
    for i, ct in enumerate((a, b)):
        ctx.fill_uniform(ct.buf, B * 2 * L, primes[:L], seed=11 + i)
    ev = api.Evaluator(ctx)
    out = api.Ciphertext(ctx, B, 3, L, capacity=3)
    for _ in range(2):
        ev.multiply(a, b, out)
    ta.synchronize()
    h = int(out.cpu().view("uint64").sum(dtype="uint64"))
    lib = ctx.lib
    capi.check(lib, lib.troyhip_ktime_enable(1))
    for _ in range(6):
        ev.multiply(a, b, out)
    ta.synchronize()
    buf = C.create_string_buffer(1 << 16)
    capi.check(lib, lib.troyhip_ktime_report(buf, len(buf)))
    print("checksum", h)
    for line in buf.value.decode().splitlines():
        if "behz" in line or "total" in line.lower():
            print("   ", line)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        B = sys.argv[1] if len(sys.argv) > 1 else "32"
        for mode in ("", "mfma1", "valu"):
            env = dict(os.environ)
            if mode:
                env["TROYHIP_BEHZ"] = mode
            print("TROYHIP_BEHZ=%s" % (mode or "(default)"), flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", B], env=env, check=False)
