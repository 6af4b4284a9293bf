"""GPU parity tests (run on a real MI355X with `-m gpu`): the product path -- libtroyhip.so through the C ABI --
against (a) the golden files generated from the real reference, (b) the CPU oracle on fresh seeded inputs, and
(c) size-independent properties at the BASELINE sizes.  Everything is bit-exact (integer arithmetic)."""
import os

import numpy as np
import pytest

import cases
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import troy_amd as ta
    from troy_amd import capi
    capi.load()  # the gfx950 library or a loud failure -- never a fallback
    ta.KernelProvider.initialize(0)
    return ta


def test_native_library_is_loaded(gpu):
    from troy_amd import capi
    assert capi.load().troyhip_build_info() == b"gfx950"
    assert any("libtroyhip.so" in line for line in open("/proc/self/maps"))


@pytest.mark.parametrize("name", cases.SMALL + cases.MEDIUM + cases.LARGE)
def test_context_tables_match_reference(name, gpu, golden_params):
    cfg, gp = cases.CONFIGS[name], golden_params[name]
    be = cases.GpuBackend(cfg)
    assert [str(p) for p in be.primes] == gp["primes"] and str(be.t) == gp["plain_modulus"]
    assert [be.ctx.key_limbs - be.ctx.last_limbs + 1, be.ctx.first_limbs, be.ctx.last_limbs] == gp["chain"]
    for limbs, lv in gp["levels"].items():
        bsk, gamma = be.ctx.behz_bases(int(limbs))
        cases.check_aux_base(bsk, gamma, lv, be.primes, cfg["N"])  # internal base: the reference's under TROYHIP_AUX_BASE=reference (test below), else the library's own class
    for p in be.primes:
        t, g = be.ctx.ntt_tables(p), gp["tables"][str(p)]
        assert str(t["root"]) == g["root"] and [str(x) for x in t["inv_degree"]] == g["inv_degree"]
        for k in ("root_op", "root_quo", "inv_op", "inv_quo"):
            assert cases.sha(t[k]) == g[k]


@pytest.mark.parametrize("name", cases.SMALL + cases.MEDIUM)
def test_scenario_matches_reference(name, gpu, golden_hashes):
    cfg = cases.CONFIGS[name]
    out = cases.scenario(cases.GpuBackend(cfg, batch=3 if cfg["N"] <= 128 else 1), cfg)
    exp = golden_hashes[name]
    assert set(out) == set(exp)
    bad = [k for k, m in out.items() if cases.sha(m.data) != exp[k]["sha256"] or m.is_ntt != exp[k]["is_ntt"] or m.cf != exp[k]["cf"]
           or abs(m.scale - exp[k]["scale"]) > 1e-12 * abs(exp[k]["scale"])]
    assert not bad, bad


def test_full_limbs_n64(gpu):
    name = "bfv_n64_k3"
    full = np.load(os.path.join(GOLDEN, f"golden_full_{name}.npz"))
    out = cases.scenario(cases.GpuBackend(cases.CONFIGS[name], batch=2), cases.CONFIGS[name])
    for k in full.files:
        assert np.array_equal(out[k].data, full[k]), k


@pytest.mark.parametrize("name", cases.LARGE)
def test_headline_configs_match_reference(name, gpu, golden_hashes):
    """BFV N=2^15 L=14 multiply+relinearize (BASELINE metric) and the CKKS N=2^15 mul->relin->rescale->rotate chain"""
    cfg = cases.CONFIGS[name]
    out = cases.scenario(cases.GpuBackend(cfg), cfg, light=True)
    bad = [k for k, m in out.items() if cases.sha(m.data) != golden_hashes[name][k]["sha256"]]
    assert not bad, bad


@pytest.mark.parametrize("name,batch", [("cfgA_bfv_n4096_k3", 5), ("cfgB_bfv_n8192_k5", 3), ("ckks_n4096_k4", 4), ("bgv_n4096_k3", 4), ("bfv_n131072_k3", 1),
                                        ("bfv_n2048_k3", 3), ("bfv_n16384_k4", 2), ("ckks_n16384_k4", 2)])
def test_batched_mul_relin_vs_oracle(name, batch, gpu, oracle_lib):
    """distinct ciphertexts in every batch slot, fresh seed: product vs CPU oracle"""
    from oracle import ref
    from troy_amd import synth
    cfg = cases.CONFIGS[name]
    be = cases.GpuBackend(cfg)
    ob = cases.oracle_backend(cfg)
    L, N, ntt = len(be.primes) - 1, cfg["N"], cfg["scheme"] == cases.CKKS
    rk = synth.uniform_kswitch_key(777, be.primes, N)
    be.set_relin_key(rk)
    ob.set_relin_key(rk)
    xa, xb = synth.uniform_ct(778, be.primes[:L], 2, N, batch), synth.uniform_ct(779, be.primes[:L], 2, N, batch)
    a = gpu.Ciphertext.from_numpy(be.ctx, xa, ntt)
    b = gpu.Ciphertext.from_numpy(be.ctx, xb, ntt)
    r = be.ev.multiply(a, b)
    be.ev.relinearizeInplace(r, be.rlk)
    got = r.cpu()
    for i in range(batch):
        e = ob.relinearize(ob.multiply(ref.Ct(xa[i], ntt), ref.Ct(xb[i], ntt)))
        assert np.array_equal(got[i], e.data), i


@pytest.mark.parametrize("logn", list(range(1, 18)))
def test_ntt_every_size(logn, gpu, oracle_lib):
    from troy_amd import synth
    N = 1 << logn
    primes = gpu.CoeffModulus.Create(N, [50, 40, 60])
    ctx = gpu.SEALContext(gpu.CKKS, N, primes, 0)
    rows = 7  # ragged: not a multiple of the limb period
    x = synth.uniform_rows(logn, primes, rows, N)
    buf = gpu.DeviceBuffer.from_numpy(x)
    ctx.ntt(buf, rows, primes)
    y = buf.to_numpy().reshape(rows, N)
    for r in range(rows):
        assert np.array_equal(y[r], oracle_lib.ntt_standalone(N, primes[r % 3], x[r], 1)), r
    ctx.ntt(buf, rows, primes, inverse=True)
    assert np.array_equal(buf.to_numpy().reshape(rows, N), x)
    # inverse on its own against the oracle
    buf2 = gpu.DeviceBuffer.from_numpy(x)
    ctx.ntt(buf2, rows, primes, inverse=True)
    z = buf2.to_numpy().reshape(rows, N)
    for r in range(rows):
        assert np.array_equal(z[r], oracle_lib.ntt_standalone(N, primes[r % 3], x[r], 3)), r


def test_ntt_properties_at_full_size(gpu):
    """BASELINE size N=2^15, 15 primes, hundreds of rows: identities that need no oracle"""
    from troy_amd import synth
    cfg = cases.CONFIGS["cfgNS_bfv_n32768_k15"]
    be = cases.GpuBackend(cfg)
    N, primes = cfg["N"], be.primes
    rows = 15 * 8
    a = gpu.DeviceBuffer(rows * N)
    b = gpu.DeviceBuffer(rows * N)
    be.ctx.fill_uniform(a, rows, primes, seed=1)
    be.ctx.fill_uniform(b, rows, primes, seed=2)
    xa, xb = a.to_numpy().reshape(rows, N), b.to_numpy().reshape(rows, N)
    assert np.array_equal(xa, synth.uniform_rows(1, primes, rows, N))  # device generator == documented generator
    pcol = np.array([primes[r % 15] for r in range(rows)], dtype=np.uint64)[:, None]
    s = gpu.DeviceBuffer.from_numpy((xa + xb) % pcol)
    for buf in (a, b, s):
        be.ctx.ntt(buf, rows, primes)
    ya, yb, ys = (x.to_numpy().reshape(rows, N) for x in (a, b, s))
    assert np.array_equal((ya + yb) % pcol, ys)                       # linearity
    assert (ya < pcol).all()                                           # canonical outputs
    be.ctx.ntt(a, rows, primes, inverse=True)
    assert np.array_equal(a.to_numpy().reshape(rows, N), xa)           # round trip
    d = np.zeros((rows, N), dtype=np.uint64)
    d[:, 0] = 1
    delta = gpu.DeviceBuffer.from_numpy(d)
    be.ctx.ntt(delta, rows, primes)
    assert (delta.to_numpy() == 1).all()                               # NTT(1) = (1, ..., 1)


def test_single_pass_ntt_vs_oracle_incl_extremes(gpu, oracle_lib):
    """ntt1.hip (N = 2^15) against the oracle, row by row: uniform rows, every residue p - 1, and p - 1 / 0 alternating -- the inputs
    that drive the guard-free rounds (forward: +3p per stage; inverse: the bound doubles per stage between lite reductions) and the FP64
    rounds to their largest values -- for primes of every butterfly class (40 / 50 bits FP64, 58 bits guard-free, 60 / 61 bits guarded)"""
    from troy_amd import capi, synth
    N = 32768
    kp = gpu.CoeffModulus.Create(N, [60, 50, 58, 40, 60])
    ctx = gpu.SEALContext(gpu.BFV, N, kp, gpu.PlainModulus.Batching(N, 20))
    # four 60-bit data primes: no narrower auxiliary class fits without an extra limb, so this context keeps the reference's 61-bit primes
    kp61 = gpu.CoeffModulus.Create(N, [60] * 5)
    ctx61 = gpu.SEALContext(gpu.BFV, N, kp61, gpu.PlainModulus.Batching(N, 20))
    aux = [int(x) for x in ctx61.behz_bases(4)[0]]
    assert aux[0] >> 60 == 1, "expected the 61-bit auxiliary class at this level"
    for c, primes, fp_expected in ((ctx, kp[:4], 2), (ctx61, [aux[0], kp61[1]], 0)):
        rows = (1030 // len(primes) + 1) * len(primes)  # at least four rows per CU: the size from which the single-pass kernel is used
        x = synth.uniform_rows(77, primes, rows, N)
        for r in range(rows - 2 * len(primes), rows):
            x[r] = primes[r % len(primes)] - 1
            if r >= rows - len(primes):
                x[r, 1::2] = 0
        fp0 = capi.stat("ntt1_fp_launches")
        for mode, inverse in ((1, False), (3, True)):
            buf = gpu.DeviceBuffer.from_numpy(x)
            c.ntt(buf, rows, primes, inverse=inverse)
            y = buf.to_numpy().reshape(rows, N)
            for r in range(rows):
                assert np.array_equal(y[r], oracle_lib.ntt_standalone(N, primes[r % len(primes)], x[r], mode)), (mode, r)
        assert capi.stat("ntt1_fp_launches") == fp0 + fp_expected, "FP64 single-pass instances: the 40- and 50-bit rows take them, 60- / 61-bit rows never"


def test_single_pass_ntt_xcd_order_vs_oracle(gpu, oracle_lib):
    """a single-pass launch large enough for the XCD-aware workgroup order (ntt1.hip n1_unit: 15 primes x 200 rows -> more than two rounds of workgroups,
    a list length that is not a multiple of 8: the padded last eighth) against the oracle, EVERY row, forward and inverse"""
    from troy_amd import synth
    N = 32768
    kp = gpu.CoeffModulus.Create(N, [60] + [58] * 13 + [60])
    ctx = gpu.SEALContext(gpu.BFV, N, kp, gpu.PlainModulus.Batching(N, 20))
    rows = 15 * 201
    x = synth.uniform_rows(78, kp, rows, N)
    for mode, inverse in ((1, False), (3, True)):
        buf = gpu.DeviceBuffer.from_numpy(x)
        ctx.ntt(buf, rows, kp, inverse=inverse)
        y = buf.to_numpy().reshape(rows, N)
        for r in range(rows):
            assert np.array_equal(y[r], oracle_lib.ntt_standalone(N, kp[r % 15], x[r], mode)), (mode, r)


def test_wide_strided_pass_at_a_large_batch(gpu):
    """ntt2.hip: the tensor's first pass takes the WIDE strided form (512 threads, 64 columns x 64 rows) by itself once the launch fills the chip twice over.
    64 pairs at the headline parameters: the path counter says it ran, and items 0, 31 and 63 of the batch equal the same pairs multiplied one at a time
    (a launch that small takes the narrow form, which the reference hashes pin)."""
    from troy_amd import api, capi, synth
    cfg = cases.CONFIGS["cfgNS_bfv_n32768_k15"]
    be = cases.GpuBackend(cfg)
    L, N, B = len(be.primes) - 1, cfg["N"], 64
    xa, xb = synth.uniform_ct(901, be.primes[:L], 2, N, B), synth.uniform_ct(902, be.primes[:L], 2, N, B)
    w0 = capi.stat("ntt2_wide_launches")
    big = be.ev.multiply(api.Ciphertext.from_numpy(be.ctx, xa, False), api.Ciphertext.from_numpy(be.ctx, xb, False)).cpu()
    assert capi.stat("ntt2_wide_launches") > w0, "the wide strided pass did not run at 64 pairs"
    w1 = capi.stat("ntt2_wide_launches")
    for i in (0, 31, 63):
        one = be.ev.multiply(api.Ciphertext.from_numpy(be.ctx, xa[i:i + 1], False), api.Ciphertext.from_numpy(be.ctx, xb[i:i + 1], False)).cpu()
        assert np.array_equal(np.asarray(big)[i], np.asarray(one)[0]), i
    assert capi.stat("ntt2_wide_launches") == w1, "one pair is expected to take the narrow form"


@pytest.mark.parametrize("logn", [12, 13, 14])
def test_single_pass_ntt_small_sizes_vs_oracle(logn, gpu, oracle_lib):
    """ntt1.hip at N = 2^12 .. 2^14 (ntt1s_*: the whole limb in LDS) against the oracle, row by row, at a launch large enough for the dispatcher to
    take it by itself: uniform rows and the extreme rows of test_single_pass_ntt_vs_oracle_incl_extremes, primes of every butterfly class"""
    from troy_amd import capi, synth
    N = 1 << logn
    kp = gpu.CoeffModulus.Create(N, [60, 50, 58, 40, 60])
    ctx = gpu.SEALContext(gpu.BFV, N, kp, gpu.PlainModulus.Batching(N, 20))
    kp61 = gpu.CoeffModulus.Create(N, [60] * 5)
    ctx61 = gpu.SEALContext(gpu.BFV, N, kp61, gpu.PlainModulus.Batching(N, 20))
    aux = [int(x) for x in ctx61.behz_bases(4)[0]]
    assert aux[0] >> 60 == 1
    need = 4 * 256 * {12: 4, 13: 2, 14: 1}[logn] + 8  # four rows per workgroup slot of the chip
    for c, primes, fp_expected in ((ctx, kp[:4], 2), (ctx61, [aux[0], kp61[1]], 0)):
        rows = (need // len(primes) + 1) * len(primes)
        x = synth.uniform_rows(78, primes, rows, N)
        for r in range(rows - 2 * len(primes), rows):
            x[r] = primes[r % len(primes)] - 1
            if r >= rows - len(primes):
                x[r, 1::2] = 0
        fp0, i0 = capi.stat("ntt1_fp_launches"), capi.stat("ntt1_int_launches")
        for mode, inverse in ((1, False), (3, True)):
            buf = gpu.DeviceBuffer.from_numpy(x)
            c.ntt(buf, rows, primes, inverse=inverse)
            y = buf.to_numpy().reshape(rows, N)
            for r in list(range(0, rows, 37)) + list(range(rows - 2 * len(primes), rows)):  # a sample of the uniform rows, every extreme row
                assert np.array_equal(y[r], oracle_lib.ntt_standalone(N, primes[r % len(primes)], x[r], mode)), (mode, r)
            back = gpu.DeviceBuffer.from_numpy(y)
            c.ntt(back, rows, primes, inverse=not inverse)
            assert np.array_equal(back.to_numpy().reshape(rows, N), x), mode  # every row: the round trip
        assert capi.stat("ntt1_fp_launches") == fp0 + 2 * fp_expected and capi.stat("ntt1_int_launches") > i0, "the single-pass kernels took these launches"


def test_cfgA_add_on_device(gpu):
    f = np.load(os.path.join(GOLDEN, "cfgA_bfv_n4096_k3.npz"))
    cfg = cases.CONFIGS["cfgA_bfv_n4096_k3"]
    be = cases.GpuBackend(cfg)
    assert be.primes == [int(x) for x in f["primes"]]
    a = gpu.Ciphertext.from_numpy(be.ctx, f["ct1"])
    be.ev.addInplace(a, gpu.Ciphertext.from_numpy(be.ctx, f["ct2"]))
    assert cases.sha(a.cpu()[0]) == str(f["sum_sha256"])


@pytest.mark.parametrize("nm", ["bfv", "bgv", "ckks"])
def test_realkey_chain(nm, gpu, oracle_lib):
    """keys made by the reference's KeyGenerator: multiply -> relinearize (-> rescale) -> rotate; bit-exact, and the
    result decrypts (CPU oracle decryptor) to the expected plaintext"""
    from oracle import oracle
    f = np.load(os.path.join(GOLDEN, f"realkey_{nm}.npz"))
    scheme = dict(bfv=gpu.BFV, bgv=gpu.BGV, ckks=gpu.CKKS)[nm]
    primes, t = [int(x) for x in f["primes"]], int(f["t"])
    ctx = gpu.SEALContext(scheme, 128, primes, t)
    ev = gpu.Evaluator(ctx)
    rlk, gk = gpu.RelinKeys(ctx), gpu.GaloisKeys(ctx)
    rlk.set(0, f["relin_key"])
    gk.set_elt(int(f["galois_elt"]), f["galois_key"])
    if nm == "ckks":
        a = gpu.Ciphertext.from_numpy(ctx, f["ct1"], True, float(f["in_scale"]))
        b = gpu.Ciphertext.from_numpy(ctx, f["ct2"], True, float(f["in_scale"]))
        r = ev.multiply(a, b)
        ev.relinearizeInplace(r, rlk)
        r = ev.rescaleToNext(r)
        ev.rotateVectorInplace(r, 1, gk)
        assert np.array_equal(r.cpu()[0], f["result"]) and r.scale == float(f["result_scale"])
    else:
        a = gpu.Ciphertext.from_numpy(ctx, f["ct1"], False, 1.0, int(f["ct1_cf"]))
        b = gpu.Ciphertext.from_numpy(ctx, f["ct2"], False, 1.0, int(f["ct2_cf"]))
        r = ev.multiply(a, b)
        ev.relinearizeInplace(r, rlk)
        ev.rotateRowsInplace(r, 1, gk)
        assert np.array_equal(r.cpu()[0], f["result"]) and r.correction_factor == int(f["result_cf"])
        from oracle import ref
        O = oracle.Oracle(scheme, 128, primes, t)
        dec = O.decrypt(ref.Ct(r.cpu()[0], False, 1.0, r.correction_factor), f["secret_key"])
        assert np.array_equal(dec, f["decrypted"])


def test_edge_values(gpu, oracle_lib):
    """all-zero and all-(p-1) ciphertexts through multiply+relinearize"""
    from oracle import ref
    from troy_amd import synth
    cfg = cases.CONFIGS["bfv_n128_k4"]
    be, ob = cases.GpuBackend(cfg), cases.oracle_backend(cfg)
    rk = synth.uniform_kswitch_key(5, be.primes, 128)
    be.set_relin_key(rk)
    ob.set_relin_key(rk)
    q = be.primes[:3]
    zero = np.zeros((2, 3, 128), dtype=np.uint64)
    top = np.stack([np.stack([np.full(128, p - 1, dtype=np.uint64) for p in q])] * 2)
    for x, y in ((zero, top), (top, top), (zero, zero)):
        r = be.ev.multiply(be.ct(x, False), be.ct(y, False))
        be.ev.relinearizeInplace(r, be.rlk)
        e = ob.relinearize(ob.multiply(ref.Ct(x), ref.Ct(y)))
        assert np.array_equal(r.cpu()[0], e.data)


def test_error_conventions(gpu):
    """std::invalid_argument / std::logic_error cases of the reference (evaluator_cuda.cu:285-286, 1174-1188, 2040-2060)"""
    from troy_amd import capi, synth
    cfg = cases.CONFIGS["bfv_n128_k4"]
    be = cases.GpuBackend(cfg)
    q3, q2 = be.primes[:3], be.primes[:2]
    a3 = gpu.Ciphertext.from_numpy(be.ctx, synth.uniform_ct(1, q3, 2, 128))
    a2 = gpu.Ciphertext.from_numpy(be.ctx, synth.uniform_ct(1, q2, 2, 128))
    with pytest.raises(capi.InvalidArgument):
        be.ev.addInplace(a3.copy(), a2)                      # level mismatch
    ntt = gpu.Ciphertext.from_numpy(be.ctx, synth.uniform_ct(1, q3, 2, 128), True)
    with pytest.raises(capi.InvalidArgument):
        be.ev.multiply(ntt, ntt)                             # BFV operands cannot be in NTT form
    with pytest.raises(capi.InvalidArgument):
        be.ev.addInplace(a3.copy(), ntt)                     # NTT form mismatch
    m = be.ev.multiply(a3, a3)
    with pytest.raises(capi.InvalidArgument):
        be.ev.relinearizeInplace(m, be.rlk)                  # no relinearization keys
    with pytest.raises(capi.InvalidArgument):
        be.ev.applyGaloisInplace(a3.copy(), 3, be.gk)        # Galois key not present
    with pytest.raises(capi.LogicError):
        be.ev.rotateVectorInplace(a3.copy(), 1, be.gk)       # unsupported scheme
    a1 = gpu.Ciphertext.from_numpy(be.ctx, synth.uniform_ct(1, be.primes[:1], 2, 128))
    with pytest.raises(capi.InvalidArgument):
        be.ev.modSwitchToNext(a1)                            # end of the chain
    with pytest.raises(capi.InvalidArgument):
        gpu.SEALContext(gpu.BFV, 128, [be.primes[0], be.primes[0]], be.t)  # duplicate primes
    with pytest.raises(capi.InvalidArgument):
        gpu.SEALContext(gpu.BFV, 100, be.primes, be.t)       # N not a power of two


@pytest.mark.parametrize("scheme,bits,tbits", [(1, [40, 40, 40, 40], 10), (3, [40, 36, 36, 40], 10)])
def test_end_to_end_own_keys(scheme, bits, tbits, gpu):
    """the whole chain with the product's own keys: host keygen/encrypt -> GPU multiply, relinearize, mod-switch, rotate ->
    host decrypt == plaintext arithmetic (negacyclic product, automorphism)"""
    from test_hostcrypto import negacyclic_mul
    from oracle import oracle
    N = 128
    primes = gpu.CoeffModulus.Create(N, bits)
    t = gpu.PlainModulus.Batching(N, tbits)
    ctx = gpu.SEALContext(scheme, N, primes, t)
    kg = gpu.KeyGenerator(ctx, seed=(21, 22))
    enc, dec = gpu.Encryptor(ctx, kg.createPublicKey()), gpu.Decryptor(ctx, kg.secretKey())
    rlk, gk = gpu.RelinKeys(ctx), gpu.GaloisKeys(ctx)
    rlk.set(0, kg.createRelinKeys())
    g = ctx.galois_elt_from_step(1)
    gk.set_elt(g, kg.createGaloisKeys([g])[g])
    rng = np.random.default_rng(9)
    B = 3
    m1 = rng.integers(0, t, (B, N), dtype=np.uint64)
    m2 = rng.integers(0, t, (B, N), dtype=np.uint64)
    a = gpu.Ciphertext.from_numpy(ctx, np.stack([enc.encrypt(m1[b]) for b in range(B)]))
    b = gpu.Ciphertext.from_numpy(ctx, np.stack([enc.encrypt(m2[b]) for b in range(B)]))
    ev = gpu.Evaluator(ctx)
    r = ev.multiply(a, b)
    ev.relinearizeInplace(r, rlk)
    r = ev.modSwitchToNext(r)
    ev.applyGaloisInplace(r, g, gk)
    out = r.cpu()
    for i in range(B):
        prod = negacyclic_mul(m1[i], m2[i], t)
        expect = oracle.apply_galois(N, g, t, prod)
        assert np.array_equal(dec.decrypt(out[i], correction_factor=r.correction_factor), expect), i


def test_cfgE_ckks_matmul_semantics(gpu, oracle_lib):
    """BASELINE config E at evaluator level (app/LinearHelperCKKS.cuh:227-248): for every batch row b,
    out[b][j] = sum_i multiplyPlain(a[b][i], W[i][j]) with addInplace, NTT form; the batch dimension is the shard axis"""
    from oracle import ref
    from troy_amd import synth
    cfg = cases.CONFIGS["ckks_n4096_k4"]
    be, ob = cases.GpuBackend(cfg), cases.oracle_backend(cfg)
    N, q = cfg["N"], be.primes[:3]
    B, I, J = 4, 3, 2
    a = [synth.uniform_ct(900 + i, q, 2, N, B) for i in range(I)]                    # a[i]: batch of B ciphertexts
    W = [[synth.uniform_rows(950 + i * J + j, q, 3, N) for j in range(J)] for i in range(I)]
    for j in range(J):
        acc = None
        for i in range(I):
            prod = gpu.Ciphertext.from_numpy(be.ctx, a[i], True)
            be.ev.multiplyPlainInplace(prod, gpu.DeviceBuffer.from_numpy(W[i][j]))
            if acc is None:
                acc = prod
            else:
                be.ev.addInplace(acc, prod)
        got = acc.cpu()
        for b in range(B):
            e = None
            for i in range(I):
                p = ob.multiply_plain(ref.Ct(a[i][b], True), W[i][j])
                e = p if e is None else ob.add(e, p)
            assert np.array_equal(got[b], e.data), (j, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfgA_bfv_n4096_k3", "bgv_n4096_k3", "cfgB_bfv_n8192_k5"])
def test_plain_operands_monomial_and_per_item(name, gpu, oracle_lib):
    """addPlain / subPlain / multiplyPlain (coefficient form) / transformToNtt(Plaintext) with one plaintext per
    batch row, and the one-coefficient plaintext under the CUDA evaluator's semantics; vs the CPU oracle, bit-exact"""
    cases.check_plain_monomial_and_batch(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("bfv_n128_k4", 3), ("cfgA_bfv_n4096_k3", 3), ("cfgB_bfv_n8192_k5", 3), ("bfv_n16384_k4", 2), ("cfgNS_bfv_n32768_k15", 2),
                                        ("bgv_n4096_k3", 3), ("bgv_n128_k4", 2), ("ckks_n4096_k4", 3)])
def test_dense_and_strided_multiply(name, batch, gpu):
    """the last case is the benchmark's own configuration and operand layout (dense, fused tensor pass) at full size"""
    cases.check_dense_multiply(name, batch)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bfv_n128_k4", "bfv_n128_k5_60", "bgv_n128_k4", "ckks_n128_k6", "cfgA_bfv_n4096_k3", "bgv_n4096_k3", "ckks_n4096_k4"])
def test_gpu_decrypt_vs_oracle(name, gpu):
    cases.check_gpu_decrypt(name)


@pytest.mark.gpu
def test_gpu_decrypt_end_to_end(gpu):
    """host encrypt -> GPU multiply + relinearize -> GPU decrypt == plaintext product (BFV), own keys"""
    from test_hostcrypto import negacyclic_mul
    N = 4096
    primes = gpu.CoeffModulus.Create(N, [36, 36, 37])
    t = gpu.PlainModulus.Batching(N, 20)
    ctx = gpu.SEALContext(1, N, primes, t)
    kg = gpu.KeyGenerator(ctx, seed=(5, 6))
    enc = gpu.Encryptor(ctx, kg.createPublicKey())
    rlk = gpu.RelinKeys(ctx)
    rlk.set(0, kg.createRelinKeys())
    rng = np.random.default_rng(3)
    B = 2
    m1 = rng.integers(0, t, (B, N), dtype=np.uint64)
    m2 = np.zeros((B, N), dtype=np.uint64)
    m2[:, :9] = rng.integers(0, t, (B, 9), dtype=np.uint64)
    a = gpu.Ciphertext.from_numpy(ctx, np.stack([enc.encrypt(m1[b]) for b in range(B)]))
    b = gpu.Ciphertext.from_numpy(ctx, np.stack([enc.encrypt(m2[b]) for b in range(B)]))
    ev = gpu.Evaluator(ctx)
    r = ev.multiply(a, b)
    ev.relinearizeInplace(r, rlk)
    got = ev.decrypt(r, gpu.DeviceBuffer.from_numpy(kg.secretKey()))
    for i in range(B):
        assert np.array_equal(got[i], negacyclic_mul(m2[i], m1[i], t)), i  # sparse operand first: 9 x 4096 terms


@pytest.mark.gpu
@pytest.mark.parametrize("name", cases.CHAIN)
def test_ckks_chain_depth3_every_op(name, gpu):
    """configs[2]: multiply -> relinearize -> rescale -> rotate(1) chained to depth 3, limbs after every op vs the reference (golden_chain.json)"""
    cases.check_chain(name, batch=2)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch", [("bgv_n128_k4", 5), ("ckks_n128_k6", 3), ("bgv_n4096_k3", 8), ("cfgD_bgv_n65536_k15", 8)])
def test_distinct_batch_relinearize_rotate(name, batch, gpu):
    """configs[3] shape: a batch of DISTINCT size-3 ciphertexts, relinearize + rotateRows(1), every item against the oracle"""
    cases.check_distinct_batch_relin_rotate(name, batch=batch)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bfv_n128_k4", "ckks_n128_k6", "bgv_n128_k4", "cfgA_bfv_n4096_k3", "ckks_n4096_k4"])
def test_lwe_extract_pack_limbs_vs_oracle(name, gpu):
    """extractLWE / assembleLWE / packLWECiphertexts / fieldTrace limb for limb against the oracle restatement of src/evaluator_cuda.cu:2178-2351"""
    cases.check_lwe_limbs(name, n_lwe=3, batch=2)


@pytest.mark.gpu
def test_wire_format_on_device(gpu):
    cases.check_save_load(gpu.api if hasattr(gpu, "api") else __import__("troy_amd").api)
    cases.check_save_terms(gpu.api if hasattr(gpu, "api") else __import__("troy_amd").api)


@pytest.mark.gpu
def test_device_modarith_edge_values_and_reference_kats(gpu):
    import json
    cases.check_device_modarith(__import__("troy_amd").api, json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"))))


@pytest.mark.gpu
@pytest.mark.parametrize("name", cases.SIZES)
def test_general_ciphertext_sizes(name, gpu):
    """3x2 / 3x3 / 2x3 multiply, size-3 square, relinearize 4 -> 2 and 5 -> 2 against the reference's own outputs"""
    cases.check_general_sizes(name, batch=3)


@pytest.mark.gpu
def test_general_sizes_limits(gpu):
    """destination size 16 is the limit (SEAL_CIPHERTEXT_SIZE_MAX); 9 x 8 works and matches the oracle, 9 x 9 is 'invalid size'"""
    cases.check_size_limits("bfv_n64_k3")
    cases.check_size_limits("ckks_n128_k6")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ckks_n128_k6", "bfv_n128_k4", "bgv_n128_k4", "ckks_n4096_k4", "cfgB_bfv_n8192_k5"])
def test_modswitch_as_first_op(name, gpu):
    cases.check_modswitch_as_first_op(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["bfv_n128_k4", "bgv_n128_k4", "ckks_n128_k6", "cfgA_bfv_n4096_k3", "ckks_n4096_k4"])
def test_api_compositions(name, gpu):
    cases.check_api_compositions(name)


@pytest.mark.gpu
def test_cfgE_matmul_helper_128x128(gpu):
    cases.check_ckks_matmul_helper()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", [1, 3])
def test_lwe_extract_and_pack(scheme, gpu):
    cases.check_lwe_pack(scheme=scheme)


@pytest.mark.gpu
def test_conv2d_helper(gpu):
    cases.check_ckks_conv2d_helper()
    h = cases.check_ckks_conv2d_helper(batch=1, image=(70, 66), kernel=(3, 3), channels=(1, 2))  # blocked path: 64 x 64 blocks
    assert h.blocked


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_parameter_sets(seed, gpu, oracle_lib):
    """the whole op list at every level on seeded random parameter sets (scheme, N = 256..4096, 2..6 primes of 33..60 bits -- the sizes
    the kernels branch on: one-step BEHZ reduction and guard-free butterflies from 2^33, guarded butterflies from 2^58), product vs
    CPU oracle, limb for limb"""
    cfg, n = cases.check_random_config(seed)
    assert n is None or n > 10, cfg


@pytest.mark.parametrize("scheme", [1, 3])
@pytest.mark.parametrize("N,bits", [(8192, (46, 46, 46, 48)), (4096, (44, 45, 49)), (16384, (40, 46, 46, 46, 49))])
def test_moddown_shared_first_pass_fp_bounds(scheme, N, bits, gpu, oracle_lib):
    """one ciphertext, FP64-class special prime above the FP64-class data primes: the mod-down's separately launched last passes plan from the
    bound of the shared first pass (uniform and extreme rows and keys vs the oracle; the FP64 two-pass kernels must have run)"""
    cases.check_moddown_shared_first_pass(scheme, N=N, bits=bits)


@pytest.mark.parametrize("name", ["cfgA_bfv_n4096_k3", "cfgB_bfv_n8192_k5", "bgv_n4096_k3", "ckks_n4096_k4", "cfgNS_bfv_n32768_k15", "cfgC_ckks_n32768_k15"])
def test_relinearize_out_of_place(name, gpu):
    """the destination form of relinearize (operand read in place from size 3, every mod-down epilogue accumulating onto (c0, c1) of the
    operand) == copy + relinearizeInplace"""
    if name not in cases.CONFIGS:
        pytest.skip("configuration not in this build of the fixtures")
    cases.check_relinearize_out_of_place(name)


def test_multiply_plain_accumulate(gpu):
    """the fused sum of ciphertext x plaintext products == the multiplyPlain + addInplace loop, limb for limb"""
    cases.check_multiply_plain_accumulate()
    cases.check_multiply_plain_accumulate(N=8192, batch=5)


@pytest.mark.parametrize("seed", list(range(201, 225)))
def test_random_parameter_sets_tiny_rings(seed, gpu, oracle_lib):
    """the same at N = 2 .. 64, where every tile of every kernel is larger than a polynomial (the scenario keeps the rotations that exist there:
    |step| < N / 2); tools/tiny_soak.py is the long form (profiles/r03_random_soak.txt)"""
    cfg, n = cases.check_random_config(seed, sizes=(2, 4, 8, 16, 32, 64), batch=3)
    assert n is None or n > 5, cfg


@pytest.mark.parametrize("seed", list(range(101, 109)))
def test_random_parameter_sets_large(seed, gpu, oracle_lib):
    """the same at N = 8192 .. 32768 (multiply, relinearize, rotate / rescale at the first level): the two-pass transform with its fused
    key-switch and tensor passes, and the small-base BEHZ kernels, on prime sizes no fixed configuration has"""
    cfg, n = cases.check_random_config(seed, sizes=(8192, 16384, 32768), batch=1, light=True)
    assert n is None or n >= 3, cfg


@pytest.mark.parametrize("big", [False, True, "small"])
@pytest.mark.parametrize("K", list(range(2, 19)))
def test_bfv_multiply_every_limb_count(K, big, gpu, oracle_lib):
    """BEHZ kernels at L = 1 .. 17: every k-block count of the matrix-core form, 30/32-, 40/45- and 55/60-bit primes (4 to 8 digit rows
    per output, the two-word reduction below 2^33), extreme residues, VALU kernels past 15 limbs"""
    cases.check_bfv_multiply_limb_count(K, big=big)


def test_behz_kernel_family_by_base(gpu, oracle_lib):
    """which BEHZ kernels a base gets (path counters, troyhip_stat): small bases of narrow primes the register-resident FP64 form (behz3.hip), up to
    15 limbs the matrix cores (behz2.hip), beyond that the VALU kernels (behz.hip) -- each against the oracle in check_bfv_multiply_limb_count"""
    from troy_amd import capi
    names = ("behz_fp_launches", "behz_mfma_launches", "behz_valu_launches")
    for K, big, expect in ((3, False, 0), (5, False, 0), (7, False, 0), (5, True, 1), (9, False, 1), (16, True, 1), (18, False, 2)):
        before = [capi.stat(n) for n in names]
        cases.check_bfv_multiply_limb_count(K, big=big)
        delta = [capi.stat(n) - b for n, b in zip(names, before)]
        assert delta[expect] >= 2 and sum(delta) == delta[expect], (K, big, delta)


PROBES_LIB = os.path.join(ROOT, "tools", "probe_libs", "libtroyhip_probes.so")  # `make -C troy_amd/csrc probes` (__graft_entry__.build() makes it)


def _hashes_in_child(names, env, timeout=900):
    """cases.mul_relin_hash of `names` in a child process under `env` (the library reads its switches once per process)"""
    import subprocess
    import sys
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import troy_amd as ta, cases\n"
            "ta.KernelProvider.initialize(0)\n"
            "print(' '.join(cases.mul_relin_hash(n) for n in %r))\n") % (tests_dir, os.path.dirname(tests_dir), names)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.split()[-len(names):]


SMALL_NAMES = ["cfgA_bfv_n4096_k3", "cfgB_bfv_n8192_k5", "bgv_n4096_k3", "ckks_n4096_k4", "bfv_n16384_k4"]
HEADLINE_NAMES = ["cfgNS_bfv_n32768_k15", "cfgC_ckks_n32768_k15"]


@pytest.mark.parametrize("env", [{"TROYHIP_NTT": "single"}, {"TROYHIP_NTT": "twopass"}, {"TROYHIP_FP64": "off"}, {"TROYHIP_AUX_BASE": "reference"},
                                 {"TROYHIP_SMALL": "split"}, {"TROYHIP_SMALL": "merged"}, {"TROYHIP_SMALL": "merged", "TROYHIP_FP64": "off"},
                                 {"TROYHIP_NTT": "single", "TROYHIP_FP64": "off"}])
def test_library_switches_agree(env, gpu):
    """the FOUR switches the shipped library reads (rt.h: TROYHIP_NTT, TROYHIP_FP64, TROYHIP_AUX_BASE, TROYHIP_SMALL; read once per process): single-pass
    against two-pass transforms at N = 2^12 .. 2^14, integer against FP64 instances, the reference's auxiliary base against the library's own, the
    merged forms of small launches against the per-base kernels -- the same limbs through multiply, relinearize (rescale) and a rotation as the
    default path, which the tests above pin against the oracle and the golden files"""
    assert _hashes_in_child(SMALL_NAMES, env) == [cases.mul_relin_hash(n) for n in SMALL_NAMES]


@pytest.mark.parametrize("env", [{"TROYHIP_NTT": "single"}, {"TROYHIP_NTT": "twopass"}, {"TROYHIP_FP64": "off"}, {"TROYHIP_AUX_BASE": "reference"},
                                 {"TROYHIP_SMALL": "split"}, {"TROYHIP_SMALL": "merged"}])
def test_library_switches_agree_at_headline_size(env, gpu):
    """the same at N = 2^15 (BFV headline parameters and the CKKS chain's): the single-pass transform forced at a small batch (by default it takes
    launches of four rows per CU and more), the two-pass transform forced, ..."""
    names = [n for n in HEADLINE_NAMES if n in cases.CONFIGS]
    assert names and _hashes_in_child(names, env) == [cases.mul_relin_hash(n) for n in names]


@pytest.mark.skipif(not os.path.exists(PROBES_LIB), reason="tools/probe_libs/libtroyhip_probes.so: make -C troy_amd/csrc probes")
@pytest.mark.parametrize("env", [{"TROYHIP_KS": "split"}, {"TROYHIP_TENSOR": "split"}, {"TROYHIP_BEHZ": "valu"}, {"TROYHIP_MODDOWN": "split"},
                                 {"TROYHIP_AUX_BASE": "reference", "TROYHIP_BEHZ": "valu"}, {"TROYHIP_BFLY": "guarded"}, {"TROYHIP_NTT2_MD_ORDER": "0"}])
def test_probe_build_fallback_forms_agree(env, gpu):
    """The unfused kernels are the library's fallback for the shapes the fused ones do not take (N < 4096, N = 2^17, other ciphertext sizes, more
    than 15 limbs: all exercised by the golden scenarios).  The PROBE build (-DTROYHIP_PROBES) can force them at a fused shape: the unfused key-switch
    inner product, the unfused tensor, the VALU BEHZ kernels, the element-wise BFV / BGV mod-down, guarded butterflies everywhere -- same limbs as
    the shipped library's default path"""
    assert _hashes_in_child(SMALL_NAMES, {**env, "TROYHIP_LIB": PROBES_LIB}) == [cases.mul_relin_hash(n) for n in SMALL_NAMES]


@pytest.mark.skipif(not os.path.exists(PROBES_LIB), reason="tools/probe_libs/libtroyhip_probes.so: make -C troy_amd/csrc probes")
@pytest.mark.parametrize("env", [{"TROYHIP_BFLY": "guarded"}, {"TROYHIP_NTT": "single", "TROYHIP_BFLY": "guarded"}, {"TROYHIP_NTT": "single", "TROYHIP_MODDOWN": "split"},
                                 {"TROYHIP_NTT": "single", "TROYHIP_CORR": "split"}, {"TROYHIP_NTT": "single", "TROYHIP_NTT1_XCD": "1"},
                                 {"TROYHIP_NTT": "single", "TROYHIP_NTT1_XCD": "1", "TROYHIP_NTT1_RPW": "1"}, {"TROYHIP_NTT": "single", "TROYHIP_NTT1_XCD": "0"},
                                 {"TROYHIP_NTT": "single", "TROYHIP_NTT1_XCD": "1", "TROYHIP_NTT1_XCD_GROUP": "3"}, {"TROYHIP_NTT2_WIDE": "1"},
                                 {"TROYHIP_NTT2_WIDE": "1", "TROYHIP_NTT": "twopass"}, {"TROYHIP_NTT2_WIDE": "0"},
                                 {"TROYHIP_NTT2_MD_ORDER": "0", "TROYHIP_NTT": "twopass"}])
def test_probe_build_fallback_forms_agree_at_headline_size(env, gpu):
    """N = 2^15 on the probe build: guarded butterflies instead of the guard-free ones, the BFV mod-down in its own kernel instead of in the inverse
    transform's epilogue, the CKKS divide-and-round correction as element-wise kernels instead of inside the forward transform, the XCD-aware
    workgroup order of the single-pass kernels (ntt1.hip n1_unit: by default only grids of two rounds of workgroups and more) forced on and off, and its
    grouped list (the mod-down and divide-and-round forms: several primes of the same rows back to back on one XCD) with a group size that leaves a short last group,
    the wide form of the forward strided pass (ntt2.hip n2_wide: by default only launches that fill the chip twice over) forced on and off"""
    names = [n for n in HEADLINE_NAMES if n in cases.CONFIGS]
    assert names and _hashes_in_child(names, {**env, "TROYHIP_LIB": PROBES_LIB}) == [cases.mul_relin_hash(n) for n in names]


def _full_batch_child(args, env, timeout=1500):
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "full_batch.py")] + args, env={**os.environ, **env}, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("[")][-1])


FALLBACK_ENV = {"TROYHIP_LIB": PROBES_LIB, "TROYHIP_KS": "split", "TROYHIP_TENSOR": "split", "TROYHIP_MODDOWN": "split", "TROYHIP_NTT": "twopass",
                "TROYHIP_NTT1_XCD": "0", "TROYHIP_NTT2_WIDE": "0", "TROYHIP_NTT2_MD_ORDER": "0"}


@pytest.mark.skipif(not os.path.exists(PROBES_LIB), reason="tools/probe_libs/libtroyhip_probes.so: make -C troy_amd/csrc probes")
def test_full_headline_batch_every_item(gpu):
    """The FULL headline batch -- BFV N = 2^15, K = 15, two lanes of 128 ciphertext pairs on two HIP streams, multiply + relinearize, as bench.py runs it:
    the wide strided pass, the XCD-aware and grouped workgroup orders of the single-pass kernels (n1_unit), the slot-fastest mod-down and the lane interleave
    exist only at this size.  EVERY one of the 256 results is compared (a 128-bit digest per item) with the same pair computed by the PROBE build in a child
    process under its fallback switches -- unfused key-switch inner product, unfused tensor, element-wise mod-down, two-pass transforms in flat workgroup
    order, narrow strided pass -- eight pairs at a time on one stream; and 16 items spread evenly over both lanes (every XCD's eighth of the unit lists) are
    compared limb for limb with the CPU oracle."""
    import full_batch
    from troy_amd import capi
    b = full_batch.Batch()
    wide0, n1 = capi.stat("ntt2_wide_launches"), capi.stat("ntt1_int_launches")
    got = b.lanes(keep=True)
    assert capi.stat("ntt2_wide_launches") > wide0 and capi.stat("ntt1_int_launches") > n1, "the full batch is expected to take the wide strided pass and the single-pass inverse"
    assert len(got) == full_batch.TOTAL and len(set(got)) == full_batch.TOTAL, "256 distinct pairs give 256 distinct results"
    picks = sorted({(i * (full_batch.TOTAL - 1)) // 15 for i in range(16)})
    assert full_batch.oracle_items(b, picks) == [], "items differ from the oracle"
    ref = _full_batch_child(["chunks", "8"], FALLBACK_ENV)
    bad = [i for i in range(full_batch.TOTAL) if got[i] != ref[i]]
    assert bad == [], f"items {bad[:16]} of the full batch differ from the fallback forms"


@pytest.mark.skipif(not os.path.exists(PROBES_LIB), reason="tools/probe_libs/libtroyhip_probes.so: make -C troy_amd/csrc probes")
def test_full_headline_batch_catches_a_perturbed_xcd_order(gpu):
    """The check above must FAIL when the XCD-aware workgroup order is wrong: the probe build can perturb n1_unit (TROYHIP_NTT1_XCD_PERTURB: one interior
    chunk of the second prime -- of the second prime group in the grouped lists -- is mapped onto its neighbour: in range, so nothing faults, but its rows are
    never transformed).  Only a few interior items change (a check of the first and last items of each lane stays green); the full comparison sees them.
    (At two lanes of 128 the plain inverse after the tensor takes the XCD-aware list; TROYHIP_NTT1_XCD = 1 also gives it to the mod-down's grouped one.)"""
    import full_batch
    good = _full_batch_child(["chunks", "8"], FALLBACK_ENV)
    bent = _full_batch_child(["lanes"], {"TROYHIP_LIB": PROBES_LIB, "TROYHIP_NTT1_XCD_PERTURB": "1", "TROYHIP_NTT1_XCD": "1"})
    bad = [i for i in range(full_batch.TOTAL) if good[i] != bent[i]]
    assert bad, "a perturbed workgroup order went unnoticed"
    half = full_batch.TOTAL // 2
    assert all(i % half not in (0, half - 1) for i in bad), bad  # interior items only: first and last of a lane are right


def _run_bench(args, timeout=900):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_rccl_single_rank(gpu):
    """the distributed code path of bench.py (RCCL barrier / reductions / all_gather) on one rank"""
    line = _run_bench(["--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--batch", "4", "--workload", "bfv_n8192_l4", "--no-cpu-baseline", "--ntt-reps", "1"])
    assert line["n_gpus"] == 1 and line["ranks"] == 1 and line["config"]["rendezvous"] == "nccl" and len(line["per_rank_ops_per_s"]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 8, 64])
def test_bench_accounts_for_every_kernel_at_small_batches(batch, gpu):
    """the library picks the single-pass inverse per launch (at least four rows per CU): at a small lane batch the key switch's accumulators take the
    two-pass kernels while the multiply's rows still run single-pass.  bench.py's byte table follows that rule -- every kernel of the step has its
    fraction, the line verifies, and nothing raises (found with `--batch 64` under two ranks)"""
    line = _run_bench(["--steps", "1", "--warmup", "0", "--batch", str(batch), "--no-cpu-baseline", "--ntt-reps", "1"])
    assert line["verified"] is True and line["config"]["batch_per_gpu"] == batch
    assert all(k.get("frac") for k in line["roofline"]["per_kernel"]), [k["name"] for k in line["roofline"]["per_kernel"] if not k.get("frac")]
    # round 6: the headline fraction is the in-step one (the standalone pair keeps its numbers beside it), the operation's own HBM rate and the
    # device's clock / power over the timed region are on the line
    roof = line["roofline"]
    assert roof["frac"] == roof["in_step"]["weighted_frac"] and roof["standalone"]["frac"] > 0 and roof["standalone"]["limb_transforms_per_launch"] > 0
    assert roof["step"]["algorithmic_bytes_per_step"] > 0 and roof["step"]["algorithmic_GBps"] > 0 and "achieved_GBps" in roof["step"]
    dev = line["rank_devices"][0]
    assert "samples" in dev and ("sclk_mhz" in dev or "note" in dev), dev


@pytest.mark.gpu
@pytest.mark.parametrize("workload,batch", [("bfv_n8192_l4", 1), ("bfv_n8192_l4", 8), ("bgv_n65536_relin_rot", 1)])
def test_bench_accounts_for_every_kernel_of_one_ciphertext(workload, batch, gpu):
    """one ciphertext (a few small ones) on one stream: the merged small-launch forms -- the byte table names the kernels that ran (a renamed
    instance would raise) and the line verifies"""
    line = _run_bench(["--steps", "1", "--warmup", "0", "--batch", str(batch), "--streams", "1", "--workload", workload, "--no-cpu-baseline", "--ntt-reps", "1"])
    assert line["verified"] is True and line["config"]["batch_per_gpu"] == batch
    assert all(k.get("frac") for k in line["roofline"]["per_kernel"]), [k["name"] for k in line["roofline"]["per_kernel"] if not k.get("frac")]


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo(gpu):
    """the N > 1 path of bench.py with real device work on a one-GPU box: two ranks under torch.distributed.run share GPU 0 and rendezvous over gloo
    (`--allow-gloo`, development only) -- barrier, MAX / MIN / SUM reductions, per-rank rates and placements, one JSON line from rank 0"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--allow-gloo", "--batch", "16", "--workload", "bfv_n8192_l4",
                        "--no-cpu-baseline", "--ntt-reps", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["config"]["rendezvous"] == "gloo" and len(line["per_rank_ops_per_s"]) == 2
    assert line["verified"] is True and len(line["rank_devices"]) == 2
    assert abs(line["value"] - sum(line["per_rank_ops_per_s"])) / line["value"] < 0.2


@pytest.mark.gpu
def test_bench_eight_ranks_share_one_gpu_over_gloo(gpu):
    """the launch the driver makes on the 8-GPU node, on a one-GPU box: EIGHT ranks under torch.distributed.run share GPU 0 (gloo rendezvous,
    `--allow-gloo`): rank -> device mapping, eight contexts and key uploads, all_gather_object of the placements, the barrier and the MAX / SUM
    reductions, ONE JSON line with ranks == 8 (round-4 verdict, item 8: no 8-GPU node has been available in five rounds)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29547",
                        os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--allow-gloo", "--batch", "8", "--workload", "bfv_n8192_l4",
                        "--no-cpu-baseline", "--ntt-reps", "1"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks"] == 8 and line["config"]["rendezvous"] == "gloo" and len(line["per_rank_ops_per_s"]) == 8
    assert line["verified"] is True and len(line["rank_devices"]) == 8 and line["scaling"] == "weak"
    # value = the units of ALL ranks over the time of the SLOWEST (the contract's max over ranks) = 8 x the smallest per-rank rate; eight processes taking
    # turns on one GPU finish far apart, so the sum of the per-rank rates is not that
    assert line["config"]["batch_per_gpu"] == 8 and abs(line["value"] - 8 * min(line["per_rank_ops_per_s"])) / line["value"] < 0.02
    assert line["value"] <= sum(line["per_rank_ops_per_s"]) * 1.001


@pytest.mark.gpu
def test_dist_device_views_single_rank_rccl(gpu, tmp_path):
    """troy_amd/dist.py's RCCL branch on ONE rank: `_DevView` (torch.as_tensor over a library allocation through __cuda_array_interface__, no copy, writes
    visible to the library), key broadcast, scatter_batch from a host array and from a device-resident batch, gather_batch / gather_batch_device -- so the
    code the 8-GPU node will run first has run somewhere (the two-rank RCCL tests below need two devices and have never had them)"""
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "from troy_amd import api, capi, dist as tdist, synth\n"
        "torch.cuda.set_device(0)\n"
        "api.KernelProvider.initialize(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "N = 4096\n"
        "primes = api.CoeffModulus.Create(N, [40, 40, 40])\n"
        "ctx = api.SEALContext(capi.CKKS, N, primes, 0)\n"
        "host = np.arange(5000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)\n"
        "buf = api.DeviceBuffer.from_numpy(host)\n"
        "t = tdist._tensor(buf, buf.words)\n"                       # zero-copy view of the library's allocation
        "assert t.is_cuda and t.data_ptr() == buf.ptr and np.array_equal(t.cpu().numpy().view(np.uint64), host)\n"
        "t += 1\n"                                                   # a write through torch is a write to the library's buffer
        "torch.cuda.synchronize()\n"
        "assert np.array_equal(buf.to_numpy(buf.words), host + np.uint64(1))\n"
        "part = tdist._tensor(buf, 100, 40)\n"                      # a sub-range
        "assert part.data_ptr() == buf.ptr + 320 and int(part[0].item()) == int((host[40] + np.uint64(1)).view(np.int64))\n"
        "tdist.broadcast(buf)\n"
        "assert np.array_equal(buf.to_numpy(buf.words), host + np.uint64(1))\n"
        "full = synth.uniform_ct(9, primes[:2], 2, N, 5)\n"
        "mine = tdist.scatter_batch(ctx, full, 5, 2, 2, is_ntt_form=True)\n"
        "api.Evaluator(ctx).negateInplace(mine)\n"
        "out = tdist.gather_batch(mine, 5)\n"
        "p = np.array(primes[:2], dtype=np.uint64)[None, None, :, None]\n"
        "assert np.array_equal(out, np.where(full == 0, full, p - full))\n"
        "dev_full = api.Ciphertext.from_numpy(ctx, full, True)\n"
        "mine2 = tdist.scatter_batch(ctx, dev_full, 5, 2, 2, is_ntt_form=True)\n"
        "back = tdist.gather_batch_device(mine2, 5)\n"
        "assert np.array_equal(back.cpu(), full)\n"
        "strided = api.Ciphertext.from_numpy(ctx, full, True, 1.0, 1, capacity=3)\n"  # capacity > size: the shard view is compacted on the device
        "assert np.array_equal(tdist._shard_tensor(strided, 5).cpu().numpy().view(np.uint64).reshape(full.shape), full)\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "os.write(1, b'single rank rccl ok\\n')\n" % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "single rank rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_bench_self_launch_two_ranks(gpu):
    """`python bench.py --gpus 2` (no launcher): two ranks over RCCL; needs two devices"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    line = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--workload", "bfv_n8192_l4", "--no-cpu-baseline", "--ntt-reps", "1"])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["config"]["rendezvous"] == "nccl" and len(line["per_rank_ops_per_s"]) == 2


@pytest.mark.gpu
def test_dist_scatter_gather_two_ranks_rccl(gpu, tmp_path):
    """troy_amd/dist.py over RCCL on device views (no host bounce); needs two devices"""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, torch.distributed as dist\n"
        "from troy_amd import api, capi, dist as tdist, synth\n"
        "lr = int(os.environ['LOCAL_RANK']); torch.cuda.set_device(lr)\n"
        "api.KernelProvider.initialize(lr)\n"
        "dist.init_process_group('nccl')\n"
        "r = dist.get_rank()\n"
        "N = 4096\n"
        "primes = api.CoeffModulus.Create(N, [40, 40, 40])\n"
        "ctx = api.SEALContext(capi.CKKS, N, primes, 0)\n"
        "full = synth.uniform_ct(9, primes[:2], 2, N, 5) if r == 0 else None\n"
        "mine = tdist.scatter_batch(ctx, full, 5, 2, 2, is_ntt_form=True)\n"
        "api.Evaluator(ctx).negateInplace(mine)\n"
        "out = tdist.gather_batch(mine, 5)\n"
        "if r == 0:\n"
        "    p = np.array(primes[:2], dtype=np.uint64)[None, None, :, None]\n"
        "    assert np.array_equal(out, np.where(full == 0, full, p - full))\n"
        "dev_full = api.Ciphertext.from_numpy(ctx, full, True) if r == 0 else None\n"   # device-resident batch: isend from device views, irecv into the result buffer
        "mine2 = tdist.scatter_batch(ctx, dev_full, 5, 2, 2, is_ntt_form=True)\n"
        "back = tdist.gather_batch_device(mine2, 5)\n"
        "if r == 0: assert np.array_equal(back.cpu(), full)\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "os.write(1, ('rank %%d ok\\n' %% r).encode())\n" % ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", port, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


@pytest.mark.gpu
def test_context_is_owned_by_one_stream(gpu):
    """ADVICE r1 / VERDICT r1: a context's scratch arena belongs to one stream; an operation arriving on another stream is refused
    (TROYHIP_LOGIC_ERROR) until troyhip_context_release_stream -- instead of silently corrupting the first stream's scratch"""
    import ctypes as C
    from troy_amd import capi, synth
    lib = capi.load()
    N = 4096
    primes = gpu.CoeffModulus.Create(N, [40, 40, 40])
    ctx = gpu.SEALContext(gpu.BFV, N, primes, gpu.PlainModulus.Batching(N, 16))
    s1, s2 = C.c_void_p(), C.c_void_p()
    capi.check(lib, lib.troyhip_stream_create(C.byref(s1)))
    capi.check(lib, lib.troyhip_stream_create(C.byref(s2)))
    xa = synth.uniform_ct(1, primes[:2], 2, N, 2)
    a = gpu.Ciphertext.from_numpy(ctx, xa)
    b = gpu.Ciphertext.from_numpy(ctx, synth.uniform_ct(2, primes[:2], 2, N, 2))
    ev1, ev2 = gpu.Evaluator(ctx, stream=s1), gpu.Evaluator(ctx, stream=s2)
    r1 = ev1.multiply(a, b)
    with pytest.raises(capi.LogicError):
        ev2.multiply(a, b)
    gpu.synchronize(s1)
    capi.check(lib, lib.troyhip_context_release_stream(ctx.h))
    r2 = ev2.multiply(a, b)
    gpu.synchronize(s2)
    assert np.array_equal(r1.cpu(), r2.cpu())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(scheme=cases.CKKS, N=32768, bits=[60, 40, 40, 60], tbits=0), dict(scheme=cases.CKKS, N=8192, bits=[50, 40, 40], tbits=0),
                                 dict(scheme=cases.BGV, N=8192, bits=[40, 36, 36, 40], tbits=20), dict(scheme=cases.BFV, N=4096, bits=[36, 36, 37], tbits=20)])
def test_rescale_onto_itself(cfg, gpu):
    """a strided batch rescaled / mod-switched onto its own buffer through the C ABI (overlapping input and output ranges)"""
    cases.check_rescale_onto_itself(cfg, batch=5)


@pytest.mark.parametrize("N,bits,scheme", [(8192, [60, 40, 40, 40, 60], cases.CKKS), (16384, [50, 49, 50, 48, 55], cases.BGV), (65536, [60, 50, 50, 60], cases.BGV),
                                           (32768, [60, 40, 50, 30, 60], cases.CKKS), (4096, [36, 36, 37], cases.BFV), (8192, [50, 50, 50, 50], cases.CKKS)])
def test_fp64_key_switch_instances_vs_oracle(N, bits, scheme, gpu):
    """the FP64 instances of the key-switch forward pair (ntt2_fp_kernel: primes below 2^50) against the oracle, limb for limb: mixed prime
    sizes (FP64 and integer classes in one key switch, wide source digits into narrow output primes and the reverse), 50-bit primes (the bound
    walk must place reductions), all-narrow sets -- and the path counter proves the FP64 kernels are what ran"""
    from troy_amd import capi
    cfg = dict(scheme=scheme, N=N, bits=bits, tbits=0 if scheme == cases.CKKS else 20)
    before = capi.stat("ks_fp_launches")
    got = cases.scenario(cases.GpuBackend(cfg, batch=2), cfg, light=True)
    exp = cases.scenario(cases.oracle_backend(cfg), cfg, light=True)
    assert not cases.compare(got, exp)
    assert capi.stat("ks_fp_launches") > before, "the FP64 key-switch instances did not run"


@pytest.mark.parametrize("N", [4096, 32768, 65536])
@pytest.mark.parametrize("width", [34, 40, 41, 47, 48, 49, 50])
def test_fp64_key_switch_prime_widths(N, width, gpu):
    """the FP64 class of the fused key switch at every width it takes -- 34, 40, 41, 47, 48, 49 and 50-bit primes (the bound walk places its reductions
    differently at each) -- at the narrowest (N = 2^16: 16 columns), the headline's (2^15) and a wide (2^12: 256 columns) strided tile, next to a 60-bit
    prime whose rows take the integer kernels in the same key switch: relinearize, rotate and rescale against the oracle, limb for limb"""
    from troy_amd import capi
    cfg = dict(scheme=cases.CKKS, N=N, bits=[width, width, width, 60], tbits=0)
    before = capi.stat("ks_fp_launches")
    got = cases.scenario(cases.GpuBackend(cfg, batch=2), cfg, light=True)
    exp = cases.scenario(cases.oracle_backend(cfg), cfg, light=True)
    assert not cases.compare(got, exp)
    assert capi.stat("ks_fp_launches") > before, "the FP64 key-switch instances did not run"


@pytest.mark.parametrize("bits", [[50, 50, 50, 50, 50, 50, 50, 60], [40] * 7 + [60], [50, 30, 50, 30, 45, 60]])
def test_fp64_key_switch_extreme_residues(bits, gpu):
    """worst-case operands for the FP64 value bounds: every ciphertext and key residue at p - 1 (largest magnitude through every stage and
    the largest accumulator sums), at (p - 1) / 2 and alternating p - 1 / 0, relinearize and rotate at the top level against the oracle"""
    from oracle import ref as R
    from troy_amd import capi
    N = 8192
    cfg = dict(scheme=cases.BGV, N=N, bits=bits, tbits=20)
    be, orc = cases.GpuBackend(cfg, batch=3), cases.oracle_backend(cfg)
    primes = be.primes
    L, K = len(primes) - 1, len(primes)
    pk = np.array(primes, dtype=np.uint64)
    key = np.empty((L, 2, K, N), dtype=np.uint64)
    key[:] = (pk - np.uint64(1))[None, None, :, None]
    be.set_relin_key(key)
    orc.set_relin_key(key)
    pl = pk[:L]
    x = np.empty((3, 3, L, N), dtype=np.uint64)
    x[0] = (pl - np.uint64(1))[None, :, None]
    x[1] = ((pl - np.uint64(1)) // np.uint64(2))[None, :, None]
    x[2] = (pl - np.uint64(1))[None, :, None]
    x[2, :, :, 1::2] = 0
    before = capi.stat("ks_fp_launches")
    c = be.api.Ciphertext.from_numpy(be.ctx, x, False, 1.0, 1)
    be.ev.relinearizeInplace(c, be.rlk)
    got = c.cpu()
    assert capi.stat("ks_fp_launches") > before
    for b in range(3):
        want = orc.impl.eval(R.OP_RELIN, R.Ct(x[b], False)).data
        assert np.array_equal(got[b], want), (bits, b)


@pytest.mark.parametrize("logn", [12, 13, 14, 16, 17])
def test_fp64_two_pass_transform_extremes(logn, gpu, oracle_lib):
    """the FP64 instances of the plain two-pass transform (ntt2_fp_kernel, primes below 2^50) against the oracle: 50-, 49-, 40- and 36-bit primes
    next to a 60-bit one (integer class in the same call), uniform rows, every residue p - 1 and p - 1 / 0 alternating -- the inputs that drive
    the signed lazy values of the FP64 butterflies (forward: + p / 2 per stage and more; inverse: doubling per stage) to their bounds"""
    from troy_amd import capi, synth
    N = 1 << logn
    primes = gpu.CoeffModulus.Create(N, [50, 40, 60, 49, 36])
    ctx = gpu.SEALContext(gpu.CKKS, N, primes, 0)
    rows = 3 * len(primes)
    x = synth.uniform_rows(logn, primes, rows, N)
    for r in range(len(primes), rows):
        x[r] = primes[r % len(primes)] - 1
        if r >= 2 * len(primes):
            x[r, 1::2] = 0
    fp0 = capi.stat("ntt2_fp_launches")
    for mode, inverse in ((1, False), (3, True)):
        buf = gpu.DeviceBuffer.from_numpy(x)
        ctx.ntt(buf, rows, primes, inverse=inverse)
        y = buf.to_numpy().reshape(rows, N)
        for r in range(rows):
            assert np.array_equal(y[r], oracle_lib.ntt_standalone(N, primes[r % len(primes)], x[r], mode)), (mode, r)
    assert capi.stat("ntt2_fp_launches") == fp0 + 2


def test_auxiliary_base_reference_mode_matches_golden_tables(gpu, golden_params):
    """TROYHIP_AUX_BASE=reference (child process: read once): the BEHZ bases of every level of every BFV configuration equal the reference's; the
    default base's results are pinned by the golden hashes above and compared with this mode in test_unfused_kernel_paths_agree /
    test_ntt_forms_agree_at_headline_size"""
    import json
    import subprocess
    import sys
    names = [n for n in cases.SMALL + cases.MEDIUM + cases.LARGE if cases.CONFIGS[n]["scheme"] == cases.BFV]
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import troy_amd as ta, cases\n"
            "ta.KernelProvider.initialize(0)\n"
            "out = {}\n"
            "for n in %r:\n"
            "    be = cases.GpuBackend(cases.CONFIGS[n])\n"
            "    out[n] = {str(l): [[str(x) for x in be.ctx.behz_bases(l)[0]], str(be.ctx.behz_bases(l)[1])] for l in range(be.ctx.last_limbs, len(be.primes) + 1)}\n"
            "print('RESULT ' + json.dumps(out))\n") % (tests_dir, os.path.dirname(tests_dir), names)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "TROYHIP_AUX_BASE": "reference"}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for n in names:
        for limbs, lv in golden_params[n]["levels"].items():
            assert got[n][limbs] == [lv["bsk"], lv["gamma"]], (n, limbs)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme,bits", [(cases.BFV, [40, 40, 40]), (cases.CKKS, [50, 40, 50])])
def test_empty_batch_is_a_noop(scheme, bits, gpu):
    """batch = 0 (the reference has no batch: an empty one is this library's own edge): every entry point returns success, launches nothing and
    leaves operands and metadata as they were"""
    import ctypes as C
    from troy_amd import api, capi, synth
    lib = capi.load()
    N = 4096
    primes = gpu.CoeffModulus.Create(N, bits)
    ctx = gpu.SEALContext(scheme, N, primes, gpu.PlainModulus.Batching(N, 20) if scheme != cases.CKKS else 0)
    ntt = scheme == cases.CKKS
    L = len(primes) - 1
    x = synth.uniform_ct(1, primes[:L], 2, N, 2)
    a, b = (api.Ciphertext.from_numpy(ctx, x, ntt, capacity=3) for _ in range(2))
    key = api.DeviceBuffer.from_numpy(synth.uniform_kswitch_key(2, primes, N))
    zero = C.c_uint64(0)
    sa, sb, out = a.struct(), b.struct(), a.struct()
    calls = {
        "multiply": lambda: lib.troyhip_multiply(ctx.h, C.byref(sa), C.byref(sb), C.byref(out), zero, None),
        "add": lambda: lib.troyhip_add(ctx.h, C.byref(sa), C.byref(sb), zero, None),
        "sub": lambda: lib.troyhip_sub(ctx.h, C.byref(sa), C.byref(sb), zero, None),
        "negate": lambda: lib.troyhip_negate(ctx.h, C.byref(sa), zero, None),
        "apply_key_switching": lambda: lib.troyhip_apply_key_switching(ctx.h, C.byref(sa), C.c_void_p(key.ptr), zero, None),
        "negacyclic_shift": lambda: lib.troyhip_negacyclic_shift(ctx.h, C.byref(sa), C.c_uint64(3), zero, None),
    }
    for name, call in calls.items():
        assert call() == 0, (name, lib.troyhip_last_error())
    gpu.synchronize()
    assert np.array_equal(a.cpu(), x) and np.array_equal(b.cpu(), x)
    ctx.ntt(api.DeviceBuffer(8), 0, primes[:1])  # a transform of zero rows


def test_host_threads_with_own_contexts(gpu):
    """six host threads, each with its own context, evaluator and HIP stream, run multiply + relinearize + rotate of three schemes at once
    (tools/threads_probe.py; ctypes drops the GIL inside the library): every thread's limbs equal the single-threaded run -- the shared state (the
    caching device pool, the error slot, the launch timing) holds"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(cases.ROOT if hasattr(cases, "ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "threads_probe.py"), "6", "8"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
