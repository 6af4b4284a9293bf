"""the wire format of CiphertextCuda::save / saveTerms (src/ciphertext_cuda.cu:16-104), its seeded variant (:26-35, 145-190) and the key
serializers (src/publickey_cuda.cuh:252, src/secretkey_cuda.cuh:292 over src/plaintext_cuda.cu:7-14, src/kswitchkeys_cuda.cuh:330-339), byte for byte.

tests/cpp/dump_wire.cpp writes what include/troyn.hpp serializes NEXT TO the raw words of the same objects.  The expected bytes are built HERE, from the
reference's field order (struct.pack below; sizes: bool 1, size_t 8, double 8, no padding -- every field is its own stream.write), the raw words and the
reference's own parms_id (tests/golden/golden_wire.json, generated from oracle/_ref by tests/golden/gen_wire_golden.py) -- not from anything the product
computed.  A permuted payload, a wrong term order or a missing field fails the comparison.  The inverse transform an NTT-form saveTerms needs comes from
the oracle.  The CUDA half of the reference cannot be built here, so no blob written by the reference exists to compare with: parity is to its source text."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dump_wire.cpp")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_wire.json")))
N, TERMS = 64, [0, 3, 17, 63]


def _dump(tmp_path, libdir, libfile):
    exe = str(tmp_path / "dump_wire")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe, os.path.join(libdir, libfile),
                    "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    out = tmp_path / "wire"
    out.mkdir()
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    return out


def _words(path):
    return np.fromfile(str(path), dtype=np.uint64)


def _pid(tag, limbs):
    return np.array(GOLD[tag]["parms_id"][str(limbs)], dtype=np.uint64).tobytes()


def _fields(tag, limbs, ntt, size, scale, cf=1, seed=0, terms=False):
    """src/ciphertext_cuda.cu:16-25: parms_id, is_ntt_form, size, poly_modulus_degree, coeff_modulus_size, scale, correction_factor, seed, terms"""
    return _pid(tag, limbs) + struct.pack("<?QQQdQQ?", ntt, size, N, limbs, scale, cf, seed, terms)


def _ct_blob(tag, limbs, ntt, size, scale, words, seed=0):
    return _fields(tag, limbs, ntt, size, scale, 1, seed) + struct.pack("<Q", len(words)) + words.tobytes()


def _check(tag, out):
    from oracle import oracle
    ckks = tag == "ckks"
    K, first = GOLD[tag]["chain"][0], GOLD[tag]["chain"][1]
    primes = GOLD[tag]["primes"]
    scale = 2.0 ** 20 if ckks else 1.0
    read = lambda name: open(out / ("%s_%s" % (tag, name)), "rb").read()  # noqa: E731

    # the library's level ids are the reference's (every level of the chain)
    for line in read("ids.txt").decode().split("\n"):
        if line:
            f = line.split()
            assert [int(x) for x in f[1:]] == GOLD[tag]["parms_id"][f[0]], (tag, f[0])

    # ---- save: header + word count + polynomials in [poly][limb][N] order
    raw = _words(out / (tag + "_ct.raw"))
    assert raw.size == 2 * first * N
    blob = read("ct.bin")
    assert blob == _ct_blob(tag, first, ckks, 2, scale, raw)
    permuted = raw.reshape(2, first, N)[:, ::-1].reshape(-1)  # limbs swapped: the same words in another order must NOT compare equal
    assert blob != _ct_blob(tag, first, ckks, 2, scale, permuted)
    raw3 = _words(out / (tag + "_ct3.raw"))
    scale3 = float(read("ct3.scale").decode())
    assert scale3 == (scale * scale if ckks else 1.0)
    assert read("ct3.bin") == _ct_blob(tag, first, ckks, 3, scale3, raw3)

    # ---- saveTerms (src/ciphertext_cuda.cu:44-80): terms flag; in COEFFICIENT form c0[j][id] for id in termIds (outer) and limb j (inner); then the word
    # count and the words of c1 (an NTT-form ciphertext is written from its inverse transform: the oracle's, not the product's)
    coeff = raw.reshape(2, first, N).copy()
    if ckks:
        O = oracle.Oracle(oracle.CKKS, N, primes, 0)
        for p in range(2):
            for j in range(first):
                coeff[p, j] = O.ntt(j, coeff[p, j], 3)  # mode 3: inverse transform, fully reduced (oracle/troy_oracle.cpp run_ntt)
    want = _fields(tag, first, ckks, 2, scale, 1, 0, True)
    for i in TERMS:
        for j in range(first):
            want += struct.pack("<Q", int(coeff[0, j, i]))
    want += struct.pack("<Q", first * N) + coeff[1].tobytes()
    assert read("terms.bin") == want
    wrong = _fields(tag, first, ckks, 2, scale, 1, 0, True)  # limb-major term order (the transposed loop) is a different stream
    for j in range(first):
        for i in TERMS:
            wrong += struct.pack("<Q", int(coeff[0, j, i]))
    assert read("terms.bin") != wrong + struct.pack("<Q", first * N) + coeff[1].tobytes()

    # ---- the seeded form (src/ciphertext_cuda.cu:26-35): the seed field set, c0 ALONE behind it
    sym = _words(out / (tag + "_sym.raw")).reshape(2, first, N)
    seed = int(read("sym.seed").decode())
    assert seed != 0
    assert read("sym.bin") == _ct_blob(tag, first, ckks, 2, scale, sym[0].reshape(-1), seed)
    assert len(read("sym.bin")) == len(blob) - first * N * 8

    # ---- keys.  PublicKey = a size-2 NTT-form ciphertext at the key level; SecretKey = the plaintext format (parms_id, coeff_count, scale, count, words)
    pk, sk = _words(out / (tag + "_pk.raw")), _words(out / (tag + "_sk.raw"))
    assert pk.size == 2 * K * N and sk.size == K * N
    assert read("pk.bin") == _ct_blob(tag, K, True, 2, 1.0, pk)
    assert read("sk.bin") == _pid(tag, K) + struct.pack("<QdQ", sk.size, 1.0, sk.size) + sk.tobytes()
    # KSwitchKeys (src/kswitchkeys_cuda.cuh:330-339): parms_id, number of index slots, per slot the digit count and one public-key blob [2][K][N] per digit
    def kswitch(slots):
        b = _pid(tag, K) + struct.pack("<Q", len(slots))
        for key in slots:
            if key is None:
                b += struct.pack("<Q", 0)
                continue
            digits = key.reshape(K - 1, 2 * K * N)
            b += struct.pack("<Q", K - 1)
            for d in digits:
                b += _ct_blob(tag, K, True, 2, 1.0, d)
        return b
    assert read("rlk.bin") == kswitch([_words(out / (tag + "_rlk.raw"))])
    slots = [None] * N  # Galois element e lives at index (e - 1) / 2: 3 -> 1, 2N - 1 -> N - 1
    slots[1], slots[N - 1] = _words(out / (tag + "_gk1.raw")), _words(out / (tag + "_gk63.raw"))
    assert read("gk.bin") == kswitch(slots)


@pytest.mark.parametrize("tag", ["bfv", "bgv", "ckks"])
def test_wire_format_bytes_on_emulator(tag, tmp_path_factory):
    global _EMUL_OUT
    try:
        out = _EMUL_OUT
    except NameError:
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
        out = _EMUL_OUT = _dump(tmp_path_factory.mktemp("wire_emul"), os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so")
    _check(tag, out)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["bfv", "bgv", "ckks"])
def test_wire_format_bytes_on_gpu(tag, tmp_path_factory):
    global _GPU_OUT
    try:
        out = _GPU_OUT
    except NameError:
        out = _GPU_OUT = _dump(tmp_path_factory.mktemp("wire_gpu"), os.path.join(ROOT, "troy_amd"), "libtroyhip.so")
    _check(tag, out)
