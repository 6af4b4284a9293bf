"""BASELINE config A: encrypt -> add -> decrypt plumbing on the CPU, no GPU involved.
The product's host-side KeyGenerator / Encryptor / Decryptor (troy_amd/csrc/hostcrypto.cpp, C ABI troyhip_host_*) against
the reference fixture (tests/golden/cfgA_*.npz), the CPU oracle, and -- where oracle/_ref is present -- the reference's
own Encryptor / Decryptor.  Decryption must be bit-exact; fresh encryptions use our own sampler and must decrypt to the
input everywhere."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
from conftest import GOLDEN
from oracle import oracle, ref
from troy_amd import synth


@pytest.fixture(scope="module")
def ta():
    import troy_amd
    return troy_amd  # host-only contexts need neither a GPU nor KernelProvider::initialize


def negacyclic_mul(a, b, t):
    n = len(a)
    out = [0] * n
    for i, x in enumerate(a):
        if not x:
            continue
        for j, y in enumerate(b):
            k = i + j
            v = int(x) * int(y)
            if k >= n:
                out[k - n] = (out[k - n] - v) % t
            else:
                out[k] = (out[k] + v) % t
    return np.array(out, dtype=np.uint64)


def test_cfgA_decrypt_fixture_and_add(ta):
    f = np.load(os.path.join(GOLDEN, "cfgA_bfv_n4096_k3.npz"))
    primes, t = [int(x) for x in f["primes"]], int(f["t"])
    ctx = ta.SEALContext(ta.BFV, 4096, primes, t, host_only=True)
    dec = ta.Decryptor(ctx, f["secret_key"])
    assert np.array_equal(dec.decrypt(f["ct1"]), f["plain1"])
    assert np.array_equal(dec.decrypt(f["ct2"]), f["plain2"])
    q = np.array(primes[:2], dtype=np.uint64)[None, :, None]
    s = (f["ct1"] + f["ct2"]) % q                       # addInplace on the host (limb-wise modular add)
    assert cases.sha(s) == str(f["sum_sha256"])
    assert np.array_equal(dec.decrypt(s), f["decrypted"])


def test_cfgA_encrypt_with_reference_public_key(ta):
    f = np.load(os.path.join(GOLDEN, "cfgA_bfv_n4096_k3.npz"))
    primes, t = [int(x) for x in f["primes"]], int(f["t"])
    ctx = ta.SEALContext(ta.BFV, 4096, primes, t, host_only=True)
    enc = ta.Encryptor(ctx, f["public_key"], seed=(11, 12))
    c1, c2 = enc.encrypt(f["plain1"]), enc.encrypt(f["plain2"])
    assert not np.array_equal(c1, f["ct1"])              # our sampler, not the reference's PRNG
    dec = ta.Decryptor(ctx, f["secret_key"])
    assert np.array_equal(dec.decrypt(c1), f["plain1"])
    q = np.array(primes[:2], dtype=np.uint64)[None, :, None]
    assert np.array_equal(dec.decrypt((c1 + c2) % q), f["decrypted"])
    O = oracle.Oracle(oracle.BFV, 4096, primes, t)
    assert np.array_equal(O.decrypt(ref.Ct(c1), f["secret_key"]), f["plain1"])
    if ref.available():                                    # the reference's own Decryptor accepts our ciphertext
        R = ref.Ref(ref.BFV, 4096, primes, t)
        R.set_secret_key(f["secret_key"])
        d, budget = R.decrypt(ref.Ct(c1))
        assert np.array_equal(d, f["plain1"]) and budget > 20


@pytest.mark.parametrize("scheme,bits,tbits", [(1, [40, 40, 40, 40], 10), (3, [40, 36, 36, 40], 10)])
def test_own_keys_roundtrip_and_keyswitch(scheme, bits, tbits, ta):
    """own KeyGenerator: encrypt/decrypt round trip; relinearization and Galois keys drive the (oracle) evaluator and the
    results decrypt to the negacyclic product / the automorphism of the plaintext"""
    N = 128
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tbits)
    ctx = ta.SEALContext(scheme, N, primes, t, host_only=True)
    kg = ta.KeyGenerator(ctx, seed=(5, 6))
    sk, pk = kg.secretKey(), kg.createPublicKey()
    enc, dec = ta.Encryptor(ctx, pk), ta.Decryptor(ctx, sk)
    rng = np.random.default_rng(3)
    m1, m2 = rng.integers(0, t, N, dtype=np.uint64), rng.integers(0, t, N, dtype=np.uint64)
    c1, c2 = enc.encrypt(m1), enc.encrypt(m2)
    assert np.array_equal(dec.decrypt(c1), m1) and np.array_equal(dec.decrypt(c2), m2)
    O = oracle.Oracle(scheme, N, primes, t)
    assert np.array_equal(O.decrypt(ref.Ct(c1), sk), m1)
    O.set_kswitch_key(0, kg.createRelinKeys())
    g = ctx.galois_elt_from_step(1)
    O.set_kswitch_key(g, kg.createGaloisKeys([g])[g])
    prod = O.eval(ref.OP_RELIN, O.eval(ref.OP_MULTIPLY, ref.Ct(c1), ref.Ct(c2)))
    assert np.array_equal(dec.decrypt(prod.data, correction_factor=prod.correction_factor), negacyclic_mul(m1, m2, t))
    rot = O.eval(ref.OP_APPLY_GALOIS, ref.Ct(c1), iarg=g)
    expect = oracle.apply_galois(N, g, t, m1)                 # m(X) -> m(X^g) on the plaintext polynomial
    assert np.array_equal(dec.decrypt(rot.data), expect)
    if ref.available():
        R = ref.Ref(scheme, N, primes, t)
        R.set_secret_key(sk)
        assert np.array_equal(R.decrypt(ref.Ct(c1))[0], m1)   # reference Decryptor on our ciphertext + our key
        R.set_public_key(pk)
        c3 = R.encrypt(m2)                                    # reference Encryptor with OUR public key
        assert np.array_equal(dec.decrypt(c3.data, correction_factor=c3.correction_factor), m2)


@pytest.mark.parametrize("scheme,bits,tbits", [(1, [40, 40, 40, 40], 10), (3, [40, 36, 36, 40], 10)])
def test_create_keyswitching_keys(scheme, bits, tbits, ta):
    """KeyGenerator::createKeySwitchingKeys (src/keygenerator.cpp:360-366): a ciphertext under ANOTHER secret key, switched with the key
    this generator makes for it (the oracle's applyKeySwitching), decrypts under this generator's secret key"""
    N = 128
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tbits)
    ctx = ta.SEALContext(scheme, N, primes, t, host_only=True)
    mine, other = ta.KeyGenerator(ctx, seed=(5, 6)), ta.KeyGenerator(ctx, seed=(7, 8))
    assert not np.array_equal(mine.secretKey(), other.secretKey())
    ksk = mine.createKeySwitchingKeys(other.secretKey())
    assert ksk.shape == (len(primes) - 1, 2, len(primes), N)
    m = np.random.default_rng(9).integers(0, t, N, dtype=np.uint64)
    c = ta.Encryptor(ctx, other.createPublicKey(), seed=(1, 2)).encrypt(m)
    assert not np.array_equal(ta.Decryptor(ctx, mine.secretKey()).decrypt(c), m)   # not ours yet
    O = oracle.Oracle(scheme, N, primes, t)
    O.set_kswitch_key(0, ksk)
    switched = O.eval(ref.OP_APPLY_KEYSWITCH, ref.Ct(c))
    assert np.array_equal(ta.Decryptor(ctx, mine.secretKey()).decrypt(switched.data, correction_factor=switched.correction_factor), m)
    with pytest.raises(Exception):
        mine.createKeySwitchingKeys(other.secretKey()[:-1])


@pytest.mark.parametrize("scheme,bits,tbits", [(1, [40, 40, 40, 40], 10), (3, [40, 36, 36, 40], 10), (2, [40, 30, 30, 40], 0)])
def test_encrypt_zero_every_level(scheme, bits, tbits, ta):
    """Encryptor::encryptZero / encryptZeroSymmetric (src/encryptor.cpp:88-150): zero at every data level of every scheme decrypts to zero; the
    key level is refused"""
    N = 128
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tbits) if tbits else 0
    ctx = ta.SEALContext(scheme, N, primes, t, host_only=True)
    kg = ta.KeyGenerator(ctx, seed=(5, 6))
    enc, dec = ta.Encryptor(ctx, kg.createPublicKey(), seed=(3, 4)), ta.Decryptor(ctx, kg.secretKey())
    enc.setSecretKey(kg.secretKey())
    assert enc.encryptZero().shape == (2, ctx.first_limbs, N)
    for limbs in range(ctx.last_limbs, ctx.first_limbs + 1):
        for z in (enc.encryptZero(limbs), enc.encryptZeroSymmetric(limbs)):
            assert z.shape == (2, limbs, N) and z.any()
            d = dec.decrypt(z)
            if scheme == 2:  # CKKS: the RNS plaintext is the (small) noise polynomial, not zero: within a few bits of 0 modulo every prime
                for l in range(limbs):  # ... after the inverse transform (the decryption is an NTT-form RNS plaintext)
                    noise = oracle.ntt_standalone(N, primes[l], d[l], 3)
                    centred = np.minimum(noise, np.uint64(primes[l]) - noise)
                    assert int(centred.max()) < 1 << 16
            else:
                assert not d.any()
    with pytest.raises(Exception):
        enc.encryptZero(len(primes))


def test_ckks_roundtrip(ta):
    N, bits = 128, [40, 30, 30, 40]
    primes = ta.CoeffModulus.Create(N, bits)
    ctx = ta.SEALContext(ta.CKKS, N, primes, 0, host_only=True)
    kg = ta.KeyGenerator(ctx, seed=(7, 8))
    enc, dec = ta.Encryptor(ctx, kg.createPublicKey()), ta.Decryptor(ctx, kg.secretKey())
    rng = np.random.default_rng(5)
    coeffs = rng.integers(-(1 << 25), 1 << 25, N)                   # scaled plaintext polynomial, coefficient form
    limbs = 3
    plain = np.stack([oracle.ntt_standalone(N, p, np.array([int(c) % p for c in coeffs], dtype=np.uint64), 1) for p in primes[:limbs]])
    ct = enc.encrypt(plain)
    back = dec.decrypt(ct)
    for l in range(limbs):
        p = primes[l]
        got = oracle.ntt_standalone(N, p, back[l], 3).astype(object)
        centred = np.array([int(v) - p if int(v) > p // 2 else int(v) for v in got])
        assert np.max(np.abs(centred - coeffs)) < 1 << 12         # fresh-encryption noise only


def test_argument_errors(ta):
    from troy_amd import capi
    N = 128
    primes = ta.CoeffModulus.Create(N, [40])
    ctx = ta.SEALContext(ta.BFV, N, primes, ta.PlainModulus.Batching(N, 10), host_only=True)
    kg = ta.KeyGenerator(ctx)
    with pytest.raises(capi.LogicError):
        kg.createRelinKeys()                                         # K = 1: key switching unsupported (context.cpp using_keyswitching)
    with pytest.raises(capi.InvalidArgument):
        ta.Encryptor(ctx, kg.createPublicKey()).encrypt(np.zeros(N + 1, dtype=np.uint64))  # plain longer than N


def test_default_seeds_come_from_os_entropy(ta):
    """KeyGenerator(ctx) / Encryptor(ctx, pk) without a seed must not be deterministic -- the reference seeds
    its PRNG from std::random_device (src/randomgen.cpp:23,72)."""
    N = 256
    primes = ta.CoeffModulus.Create(N, [40, 41])
    t = ta.PlainModulus.Batching(N, 16)
    ctx = ta.SEALContext(ta.BFV, N, primes, t, host_only=True)
    k1, k2 = ta.KeyGenerator(ctx), ta.KeyGenerator(ctx)
    assert not np.array_equal(k1.secretKey(), k2.secretKey())
    assert np.array_equal(ta.KeyGenerator(ctx, seed=(9, 9)).secretKey(), ta.KeyGenerator(ctx, seed=(9, 9)).secretKey())  # explicit seed: test-only determinism
    m = np.arange(N, dtype=np.uint64) % np.uint64(t)
    e1, e2 = ta.Encryptor(ctx, k1.createPublicKey()), ta.Encryptor(ctx, k1.createPublicKey())
    c1, c2, c3 = e1.encrypt(m), e2.encrypt(m), e1.encrypt(m)
    assert not np.array_equal(c1, c2) and not np.array_equal(c1, c3)   # fresh (u, e0, e1) per instance and per call
    dec = ta.Decryptor(ctx, k1.secretKey())
    for c in (c1, c2, c3):
        assert np.array_equal(dec.decrypt(c), m)
    buf = (C.c_uint8 * 32)()
    from troy_amd import capi
    capi.check(ctx.lib, ctx.lib.troyhip_random_bytes(buf, 32))
    assert any(buf)


@pytest.mark.parametrize("scheme,bits,tbits", [(1, [40, 40, 40, 40], 10), (3, [40, 36, 36, 40], 10)])
def test_encrypt_symmetric_roundtrip(scheme, bits, tbits, ta):
    """Encryptor::encryptSymmetric (src/encryptor.cpp:88-148 with is_asymmetric false, src/utils/rlwe.cpp:234-345): (-(a s + e) + m, a)
    at the first level, coefficient form for BFV/BGV; decrypts with our Decryptor, the oracle and the reference's own Decryptor"""
    from troy_amd import capi
    N = 128
    primes = ta.CoeffModulus.Create(N, bits)
    t = ta.PlainModulus.Batching(N, tbits)
    ctx = ta.SEALContext(scheme, N, primes, t, host_only=True)
    kg = ta.KeyGenerator(ctx, seed=(11, 12))
    sk = kg.secretKey()
    enc, dec = ta.Encryptor(ctx, None), ta.Decryptor(ctx, sk)
    with pytest.raises(RuntimeError):
        enc.encryptSymmetric(np.zeros(N, dtype=np.uint64))          # secret key is not set
    with pytest.raises(RuntimeError):
        enc.encrypt(np.zeros(N, dtype=np.uint64))                   # public key is not set
    enc.setSecretKey(sk)
    m = np.random.default_rng(8).integers(0, t, N, dtype=np.uint64)
    c1, c2 = enc.encryptSymmetric(m), enc.encryptSymmetric(m)
    assert c1.shape == (2, ctx.first_limbs, N) and not np.array_equal(c1, c2)
    assert np.array_equal(dec.decrypt(c1), m) and np.array_equal(dec.decrypt(c2), m)
    O = oracle.Oracle(scheme, N, primes, t)
    assert np.array_equal(O.decrypt(ref.Ct(c1), sk), m)
    if ref.available():
        R = ref.Ref(scheme, N, primes, t)
        R.set_secret_key(sk)
        assert np.array_equal(R.decrypt(ref.Ct(c1))[0], m)
    with pytest.raises(capi.InvalidArgument):
        enc.encryptSymmetric(np.zeros(N + 1, dtype=np.uint64))


def test_encrypt_symmetric_ckks_every_level(ta):
    N, bits = 128, [40, 30, 30, 40]
    primes = ta.CoeffModulus.Create(N, bits)
    ctx = ta.SEALContext(ta.CKKS, N, primes, 0, host_only=True)
    kg = ta.KeyGenerator(ctx, seed=(7, 8))
    enc, dec = ta.Encryptor(ctx, None), ta.Decryptor(ctx, kg.secretKey())
    enc.setSecretKey(kg.secretKey())
    coeffs = np.random.default_rng(6).integers(-(1 << 25), 1 << 25, N)
    for limbs in (3, 2, 1):                                         # the ciphertext is sampled at the plaintext's own level
        plain = np.stack([oracle.ntt_standalone(N, p, np.array([int(c) % p for c in coeffs], dtype=np.uint64), 1) for p in primes[:limbs]])
        ct = enc.encryptSymmetric(plain)
        assert ct.shape == (2, limbs, N)
        back = dec.decrypt(ct)
        for l in range(limbs):
            p = primes[l]
            got = oracle.ntt_standalone(N, p, back[l], 3).astype(object)
            centred = np.array([int(v) - p if int(v) > p // 2 else int(v) for v in got])
            assert np.max(np.abs(centred - coeffs)) < 64           # one fresh error term (|e| <= 21 for the centred binomial)


def test_batch_encoder_against_reference_fixtures(ta):
    """BatchEncoder (src/batchencoder.cpp) pinned on what the reference itself produced: the cfgA fixture holds slot values and the plaintexts
    the reference's encoder made of them (N = 4096); the real-key fixtures hold a decrypted product whose decoding the generator checked
    against the slot-wise product rotated by one (N = 128, BFV and BGV)."""
    f = np.load(os.path.join(GOLDEN, "cfgA_bfv_n4096_k3.npz"))
    primes, t = [int(x) for x in f["primes"]], int(f["t"])
    ctx = ta.SEALContext(ta.BFV, 4096, primes, t, host_only=True)
    be = ta.BatchEncoder(ctx)
    for k in ("1", "2"):
        assert np.array_equal(be.encode(f["values" + k]), f["plain" + k])
        assert np.array_equal(be.decode(f["plain" + k]), f["values" + k])
    assert np.array_equal(be.decode(f["decrypted"]), (f["values1"] + f["values2"]) % np.uint64(t))
    short = be.encode(f["values1"][:5])                   # fewer values than slots: the rest are zero
    assert np.array_equal(be.decode(short), np.concatenate([f["values1"][:5], np.zeros(4096 - 5, dtype=np.uint64)]))
    for nm, scheme in (("bfv", ta.BFV), ("bgv", ta.BGV)):
        g = np.load(os.path.join(GOLDEN, f"realkey_{nm}.npz"))
        primes, t, N = [int(x) for x in g["primes"]], int(g["t"]), 128
        be = ta.BatchEncoder(ta.SEALContext(scheme, N, primes, t, host_only=True))
        prod = (g["values1"] * g["values2"]) % np.uint64(t)
        expect = np.concatenate([np.roll(prod[:N // 2], -1), np.roll(prod[N // 2:], -1)])
        assert np.array_equal(be.decode(g["decrypted"]), expect)
    if ref.available():                                   # and live against the reference's encoder where it is built
        rng = np.random.default_rng(5)
        R = ref.Ref(ref.BGV, 128, primes, t, seed=3)
        v = rng.integers(0, t, 128, dtype=np.uint64)
        assert np.array_equal(be.encode(v), R.batch_encode(v))
        assert np.array_equal(be.decode(R.batch_encode(v)), v)


def test_polynomial_packing_and_automorphism_keys(ta):
    """BatchEncoder::encodePolynomial / decodePolynomial (src/batchencoder_cuda.cu:124-170, 267-286) and createAutomorphismKeys
    (src/keygenerator.cpp:350-358) of the Python mirror"""
    N = 128
    primes = ta.CoeffModulus.Create(N, [40, 40, 40])
    t = ta.PlainModulus.Batching(N, 12)
    ctx = ta.SEALContext(ta.BFV, N, primes, t, host_only=True)
    be = ta.BatchEncoder(ctx)
    p = be.encodePolynomial(np.array([1, t + 5, 7], dtype=np.uint64))
    assert p.dtype == np.uint64 and list(p) == [1, 5, 7]
    assert list(be.decodePolynomial(p)) == [1, 5, 7]
    s = be.encodePolynomial(np.array([-1, 2, -3], dtype=np.int64))
    assert s.size == N and list(s[:4]) == [t - 1, 2, t - 3, 0]
    assert list(be.decodePolynomial(s, signed=True)[:4]) == [-1, 2, -3, 0]
    with pytest.raises(Exception):
        be.encodePolynomial(np.zeros(N + 1, dtype=np.uint64))
    kg = ta.KeyGenerator(ctx, seed=(5, 6))
    keys = kg.createAutomorphismKeys()
    assert sorted(keys) == [3, 5, 9, 17, 33, 65, 129] and all(k.shape == (2, 2, 3, N) for k in keys.values())
    # the key for X -> X^(N + 1) really switches sigma(s) back to s: apply the automorphism with the oracle and decrypt
    m = np.random.default_rng(1).integers(0, t, N, dtype=np.uint64)
    c = ta.Encryptor(ctx, kg.createPublicKey(), seed=(1, 2)).encrypt(m)
    O = oracle.Oracle(ta.BFV, N, primes, t)
    O.set_kswitch_key(N + 1, keys[N + 1])
    rot = O.eval(ref.OP_APPLY_GALOIS, ref.Ct(c), iarg=N + 1)
    assert np.array_equal(ta.Decryptor(ctx, kg.secretKey()).decrypt(rot.data), oracle.apply_galois(N, N + 1, t, m))
