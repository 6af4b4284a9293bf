"""Development coverage without a GPU: the SAME kernel sources (troy_amd/csrc/*.hip), compiled for the host with the
fiber SIMT emulator (tests/emul/hip_emul.h), are run against the golden files of the reference.  This exercises the
host logic (precompute, evaluator orchestration, C ABI) and the kernels' index arithmetic; it is NOT the parity claim --
that is tests/test_gpu_parity.py on real hardware -- and the emulated library is never used by the product."""
import os
import subprocess

import numpy as np
import pytest

import cases
from conftest import GOLDEN, ROOT

EMUL = os.path.join(ROOT, "tests", "emul", "libtroyhip_emul.so")


@pytest.fixture(scope="module")
def emul_api():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    from troy_amd import api, capi
    lib = capi.load(EMUL)
    old = api.KernelProvider._lib
    api.KernelProvider.initialize(0, _lib=lib)
    yield api
    api.KernelProvider._lib = old


@pytest.mark.parametrize("name", cases.SMALL + ["cfgA_bfv_n4096_k3"])
def test_emulated_kernels_match_reference(name, emul_api, golden_hashes, golden_params):
    cfg = cases.CONFIGS[name]
    be = cases.GpuBackend(cfg, batch=5 if cfg["N"] <= 128 else 1)  # 5 = one blocked group of 4 + a remainder (ks_mac)
    gp = golden_params[name]
    assert [str(p) for p in be.primes] == gp["primes"] and str(be.t) == gp["plain_modulus"]
    for limbs, lv in gp["levels"].items():
        bsk, gamma = be.ctx.behz_bases(int(limbs))
        cases.check_aux_base(bsk, gamma, lv, be.primes, cfg["N"])  # the reference's base under TROYHIP_AUX_BASE=reference, else the library's own class
    for p in be.primes:
        t, g = be.ctx.ntt_tables(p), gp["tables"][str(p)]
        for k in ("root_op", "root_quo", "inv_op", "inv_quo"):
            assert cases.sha(t[k]) == g[k]
    out = cases.scenario(be, cfg)
    exp = golden_hashes[name]
    assert set(out) == set(exp)
    for k, m in out.items():
        assert cases.sha(m.data) == exp[k]["sha256"], k
        assert m.is_ntt == exp[k]["is_ntt"] and m.cf == exp[k]["cf"]


@pytest.mark.parametrize("logn", [3, 6, 11, 12, 13, 15, 16, 17])
def test_emulated_ntt_all_sizes(logn, emul_api, oracle_lib):
    api = emul_api
    from troy_amd import synth
    N = 1 << logn
    primes = api.CoeffModulus.Create(N, [50, 40])
    ctx = api.SEALContext(api.CKKS, N, primes, 0)
    x = synth.uniform_rows(logn, primes, 4, N)
    buf = api.DeviceBuffer.from_numpy(x)
    ctx.ntt(buf, 4, primes)
    y = buf.to_numpy().reshape(4, N)
    for r in range(4):
        assert np.array_equal(y[r], oracle_lib.ntt_standalone(N, primes[r % 2], x[r], 1))
    ctx.ntt(buf, 4, primes, inverse=True)
    assert np.array_equal(buf.to_numpy().reshape(4, N), x)


@pytest.mark.parametrize("name", ["bfv_n64_k3", "bgv_n128_k4"])
def test_emulated_plain_operands(name, emul_api, oracle_lib):
    cases.check_plain_monomial_and_batch(name)


@pytest.mark.parametrize("name", ["cfgA_bfv_n4096_k3", "bgv_n4096_k3", "ckks_n4096_k4"])
def test_emulated_dense_multiply(name, emul_api):
    cases.check_dense_multiply(name, batch=2)


@pytest.mark.parametrize("name", ["bfv_n128_k4", "bgv_n128_k4", "ckks_n128_k6"])
def test_emulated_decrypt(name, emul_api):
    cases.check_gpu_decrypt(name, batch=2)


@pytest.mark.parametrize("name", ["bfv_n128_k4", "bgv_n128_k4", "ckks_n128_k6"])
def test_emulated_api_compositions(name, emul_api):
    cases.check_api_compositions(name)


@pytest.mark.parametrize("name", ["ckks_n128_k6", "bfv_n128_k4", "bgv_n128_k4"])
def test_emulated_modswitch_as_first_op(name, emul_api):
    cases.check_modswitch_as_first_op(name)


@pytest.mark.parametrize("name", ["bfv_n64_k3", "bfv_n128_k5_60", "ckks_n128_k6", "bgv_n128_k4"])
def test_emulated_general_sizes(name, emul_api):
    cases.check_general_sizes(name, batch=2)


@pytest.mark.parametrize("K,big", [(2, True), (5, False), (6, True), (9, "small"), (12, True), (15, False), (16, True), (17, True)])
def test_emulated_bfv_multiply_limb_counts(K, big, emul_api, oracle_lib):
    """both BEHZ kernels against the oracle at limb counts that cover the k-block counts of the 8-shift matrix-core form
    (behz2.hip; L = 1, 4, 5, 11, 14, 15), its two q-side reductions, extreme residues, and the VALU kernels at L = 16"""
    cases.check_bfv_multiply_limb_count(K, N=128, batch=1, big=big)


@pytest.mark.parametrize("seed", [3, 4, 5, 6, 9, 12, 15, 21])
def test_emulated_random_parameter_sets(seed, emul_api, oracle_lib):
    """the whole op list at every level on seeded random parameter sets (scheme, N, 2..6 primes of 33..60 bits -- the sizes the kernels
    branch on), product vs CPU oracle"""
    cfg, n = cases.check_random_config(seed, sizes=(64, 128, 256), batch=1)
    assert n is None or n > 10, cfg


def test_emulated_multiply_plain_accumulate(emul_api):
    cases.check_multiply_plain_accumulate(N=128, batch=2)


@pytest.mark.parametrize("scheme", [1, 3])
def test_emulated_moddown_shared_first_pass_fp_bounds(scheme, emul_api, oracle_lib):
    cases.check_moddown_shared_first_pass(scheme, N=4096, bits=(44, 45, 49))


@pytest.mark.parametrize("name", ["bfv_n64_k3", "ckks_n128_k6", "cfgA_bfv_n4096_k3", "bgv_n4096_k3", "ckks_n4096_k4"])
def test_emulated_relinearize_out_of_place(name, emul_api):
    cases.check_relinearize_out_of_place(name, batch=2)


@pytest.mark.parametrize("N,bits", [(8192, [60, 40, 40, 60]), (16384, [50, 45, 55])])
def test_emulated_ckks_two_round_strided_plans(N, bits, emul_api, oracle_lib):
    """CKKS key switch at the sizes whose strided pass has two rounds (N = 8192, 16384): the rows (digit k == output prime) are skipped by the
    first pass, which must not advance its LDS buffer alternation there -- on the emulator a violation is deterministic (on the GPU it is a race
    that the suite did not hit).  Light scenario against the oracle, limb for limb."""
    cfg = dict(scheme=cases.CKKS, N=N, bits=bits, tbits=0)
    got = cases.scenario(cases.GpuBackend(cfg, batch=1), cfg, light=True)
    exp = cases.scenario(cases.oracle_backend(cfg), cfg, light=True)
    assert not cases.compare(got, exp)


@pytest.mark.parametrize("cfg", [dict(scheme=cases.CKKS, N=128, bits=[40, 40, 40, 40], tbits=0), dict(scheme=cases.BGV, N=128, bits=[40, 36, 36, 40], tbits=10),
                                 dict(scheme=cases.CKKS, N=32768, bits=[40, 40, 40], tbits=0)])
def test_emulated_rescale_onto_itself(cfg, emul_api):
    cases.check_rescale_onto_itself(cfg, batch=2 if cfg["N"] > 4096 else 3)


def test_emulated_auxiliary_base_independence(tmp_path, golden_params, golden_hashes):
    """the BEHZ auxiliary base is internal: with the reference's 61-bit base (TROYHIP_AUX_BASE=reference) the tables equal the reference's and every
    scenario output equals the golden hashes -- the same hashes the default base (own primes of the guard-free / FP64 class) reproduces in
    test_emulated_kernels_match_reference.  Child process: the switch is read once."""
    import json
    import subprocess
    import sys
    names = ["bfv_n64_k3", "bfv_n128_k4", "bfv_n128_k5_60", "cfgA_bfv_n4096_k3"]
    script = tmp_path / "aux.py"
    script.write_text(
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from troy_amd import api, capi\n"
        "lib = capi.load(%r)\n"
        "api.KernelProvider.initialize(0, _lib=lib)\n"
        "import cases\n"
        "out = {}\n"
        "for name in %r:\n"
        "    cfg = cases.CONFIGS[name]\n"
        "    be = cases.GpuBackend(cfg, batch=1)\n"
        "    lv = {str(l): [[str(x) for x in be.ctx.behz_bases(l)[0]], str(be.ctx.behz_bases(l)[1])] for l in range(be.ctx.last_limbs, len(be.primes) + 1)}\n"
        "    out[name] = dict(levels=lv, hashes={k: cases.sha(m.data) for k, m in cases.scenario(be, cfg).items()})\n"
        "print('RESULT ' + json.dumps(out))\n" % (ROOT, os.path.join(ROOT, "tests"), EMUL, names))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, TROYHIP_AUX_BASE="reference"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    got = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for name in names:
        for limbs, lv in golden_params[name]["levels"].items():
            assert got[name]["levels"][limbs] == [lv["bsk"], lv["gamma"]], (name, limbs)
        assert got[name]["hashes"] == {k: v["sha256"] for k, v in golden_hashes[name].items()}, name


def test_emulated_size_limits(emul_api):
    cases.check_size_limits("bfv_n64_k3")
    cases.check_size_limits("ckks_n128_k6")


def test_emulated_ckks_matmul_helper(emul_api):
    cases.check_ckks_matmul_helper(N=256, bits=(40, 30, 30, 40), batch=2, dims=(24, 20))


def test_emulated_lwe_pack(emul_api):
    cases.check_lwe_pack(N=64, bits=(40, 40, 40, 40), tbits=10, n_lwe=3, batch=1)


def test_emulated_save_load_roundtrip(emul_api):
    cases.check_save_load(emul_api)


def test_emulated_save_terms_roundtrip(emul_api):
    cases.check_save_terms(emul_api)


@pytest.mark.parametrize("name", ["bfv_n128_k4", "ckks_n128_k6", "bgv_n128_k4"])
def test_emulated_lwe_limbs_vs_oracle(name, emul_api):
    cases.check_lwe_limbs(name, n_lwe=3, batch=2)


def test_emulated_ckks_chain_depth3(emul_api):
    cases.check_chain("ckks_n128_k6", batch=2)


def test_emulated_distinct_batch_relin_rotate(emul_api):
    cases.check_distinct_batch_relin_rotate("bgv_n128_k4", batch=5)
    cases.check_distinct_batch_relin_rotate("ckks_n128_k6", batch=3)


def test_emulated_device_modarith(emul_api):
    import json
    cases.check_device_modarith(emul_api, json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"))))


def test_emulated_single_pass_ntt_row_loop_and_prime_classes(oracle_lib, tmp_path):
    """ntt1.hip (N = 2^15, one HBM round trip): several limbs per workgroup (the prefetching row loop, forced by
    TROYHIP_NTT1_RPW), ragged last chunk, primes of both butterfly classes (below 2^58: guard-free; 60 / 61 bits: guarded), against
    the oracle and against the two-pass kernels.  Child processes: the row count per workgroup is read once per process."""
    import subprocess
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from troy_amd import api, capi, synth\n"
        "from oracle import oracle\n"
        "lib = capi.load(%r)\n"
        "api.KernelProvider.initialize(0, _lib=lib)\n"
        "N = 32768\n"
        "kp = api.CoeffModulus.Create(N, [60, 50, 58, 40, 60])\n"
        "ctx = api.SEALContext(api.BFV, N, kp, api.PlainModulus.Batching(N, 20))\n"
        "primes = kp[:4] + [int(ctx.behz_bases(4)[0][0])]\n"   # + a 61-bit BEHZ prime (guarded butterflies)
        "assert primes[-1] >> 60 == 1\n"
        "rows = 5 * len(primes) * 2\n"
        "x = synth.uniform_rows(7, primes, rows, N, inner=2)\n"
        "buf = api.DeviceBuffer.from_numpy(x)\n"
        "ctx.ntt(buf, rows, primes, inner=2)\n"
        "y = buf.to_numpy().reshape(rows, N)\n"
        "for r in range(rows):\n"
        "    assert np.array_equal(y[r], oracle.ntt_standalone(N, primes[(r // 2) %% len(primes)], x[r], 1)), r\n"
        "ctx.ntt(buf, rows, primes, inner=2, inverse=True)\n"
        "assert np.array_equal(buf.to_numpy().reshape(rows, N), x)\n"
        # extreme inputs of the guard-free inverse rounds (value bounds double per stage): all p - 1, and p - 1 / 0 alternating
        "rows2 = 2 * len(primes)\n"
        "e = np.zeros((rows2, N), dtype=np.uint64)\n"
        "for r in range(rows2):\n"
        "    p = primes[(r // 2) %% len(primes)]\n"
        "    e[r] = p - 1\n"
        "    if r %% 2: e[r, 1::2] = 0\n"
        "buf2 = api.DeviceBuffer.from_numpy(e)\n"
        "ctx.ntt(buf2, rows2, primes, inner=2, inverse=True)\n"
        "z = buf2.to_numpy().reshape(rows2, N)\n"
        "for r in range(rows2):\n"
        "    assert np.array_equal(z[r], oracle.ntt_standalone(N, primes[(r // 2) %% len(primes)], e[r], 3)), r\n"
        "ctx.ntt(buf2, rows2, primes, inner=2)\n"
        "assert np.array_equal(buf2.to_numpy().reshape(rows2, N), e)\n"
        "print('ok')\n" % (ROOT, os.path.join(ROOT, 'tests'), EMUL))
    for rpw, xcd in (("3", "0"), ("1", "0"), ("3", "1"), ("1", "1")):
        # small launches go to the two-pass kernels by default; the reference's auxiliary base supplies the 61-bit prime of the guarded class.
        # TROYHIP_NTT1_XCD = 1: the XCD-aware workgroup -> (prime, chunk) mapping of large grids, with its padded last eighth, at this size
        env = dict(os.environ, TROYHIP_NTT1_RPW=rpw, TROYHIP_NTT1_XCD=xcd, TROYHIP_NTT="single", TROYHIP_AUX_BASE="reference")
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_emulated_wide_strided_pass(oracle_lib, tmp_path):
    """ntt2.hip: the WIDE form of the forward strided pass (N = 2^15: 512 threads own 4096 points, 64 columns x 64 rows), which the
    library takes by itself only for launches that fill the chip twice over -- forced here (TROYHIP_NTT2_WIDE = 1, probe builds) for the plain transform of
    every prime class against the oracle, and for multiply + relinearize + rotation (the tensor's first pass) against the narrow form."""
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from troy_amd import api, capi, synth\n"
        "from oracle import oracle\n"
        "lib = capi.load(%r)\n"
        "api.KernelProvider.initialize(0, _lib=lib)\n"
        "import cases\n"
        "def wide():\n"
        "    v = ctypes.c_uint64(); lib.troyhip_stat(b'ntt2_wide_launches', ctypes.byref(v)); return v.value\n"
        "for logn in (15,):\n"
        "    N = 1 << logn\n"
        "    kp = api.CoeffModulus.Create(N, [60, 50, 58, 40, 60])\n"
        "    ctx = api.SEALContext(api.BFV, N, kp, api.PlainModulus.Batching(N, 20))\n"
        "    primes = kp[:4]\n"
        "    rows = 2 * len(primes)\n"
        "    x = synth.uniform_rows(7, primes, rows, N)\n"
        "    x[rows - 1] = primes[(rows - 1) %% len(primes)] - 1\n"
        "    buf = api.DeviceBuffer.from_numpy(x)\n"
        "    ctx.ntt(buf, rows, primes)\n"
        "    y = buf.to_numpy().reshape(rows, N)\n"
        "    for r in range(rows):\n"
        "        assert np.array_equal(y[r], oracle.ntt_standalone(N, primes[r %% len(primes)], x[r], 1)), (logn, r)\n"
        "cases.CONFIGS['w_bfv'] = {'scheme': 1, 'N': 32768, 'bits': [60, 58, 60], 'tbits': 20}\n"      # integer classes (guarded + guard-free)
        "cases.CONFIGS['w_ckks'] = {'scheme': 2, 'N': 32768, 'bits': [60, 40, 40, 60], 'tbits': 0}\n"   # FP64 class, with the rescale
        "print(cases.mul_relin_hash('w_bfv', batch=1) + '/' + cases.mul_relin_hash('w_ckks', batch=1), wide())\n"
        % (ROOT, os.path.join(ROOT, 'tests'), EMUL))
    got = {}
    for w in ("0", "1"):
        out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, TROYHIP_NTT="twopass", TROYHIP_NTT2_WIDE=w), capture_output=True, text=True, timeout=1800)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        got[w] = out.stdout.split()[-2:]
    assert got["0"][0] == got["1"][0]
    assert int(got["0"][1]) == 0 and int(got["1"][1]) >= 3, got  # the wide form really ran: the plain transform's classes, the tensors' first passes


def test_emulated_small_launch_forms_agree(tmp_path):
    """The merged forms of small launches (evaluator.cpp: both BEHZ bases through one launch per step, one first pass over the special limb and the
    data limbs of a mod-down) against the per-base kernels of the large batch, in child processes (TROYHIP_SMALL is read once): same limbs through
    multiply, relinearize and a rotation -- BFV and BGV, one ciphertext and a batch, squaring included"""
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from troy_amd import api, capi, synth\n"
        "lib = capi.load(%r)\n"
        "api.KernelProvider.initialize(0, _lib=lib)\n"
        "import cases\n"
        "out = [cases.mul_relin_hash(n, batch=b) for n in ('cfgA_bfv_n4096_k3', 'bgv_n4096_k3') for b in (1, 3)]\n"
        "cfg = cases.CONFIGS['cfgA_bfv_n4096_k3']\n"
        "be = cases.GpuBackend(cfg)\n"
        "x = synth.uniform_ct(5, be.primes[:2], 2, 4096, 2)\n"
        "c = api.Ciphertext.from_numpy(be.ctx, x)\n"
        "out.append(cases.sha(be.ev.multiply(c, c).cpu()))\n"   # squaring: one operand, extended and transformed once
        "print(' '.join(out))\n" % (ROOT, os.path.join(ROOT, 'tests'), EMUL))
    got = {}
    for mode in ("split", "merged"):
        out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, TROYHIP_SMALL=mode), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        got[mode] = out.stdout.split()[-5:]
    assert got["split"] == got["merged"] and len(got["split"]) == 5


def test_emulated_single_pass_ntt_small_sizes(oracle_lib, tmp_path):
    """ntt1.hip at N = 2^12 .. 2^14 (ntt1s_*: the whole limb in LDS, the sub-block rounds of the N = 2^15 form under a shorter cross-sub-block round):
    forward, inverse and extreme inputs for the three butterfly classes (FP64 below 2^50, guard-free below 2^58, guarded) against the oracle, with and
    without the row loop; then multiply + relinearize + rotation through the single-pass kernels (the BFV mod-down epilogue included) against the
    two-pass kernels.  Child processes: TROYHIP_NTT is read once."""
    import sys
    script = tmp_path / "w.py"
    script.write_text(
        "import sys, os, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from troy_amd import api, capi, synth\n"
        "from oracle import oracle\n"
        "lib = capi.load(%r)\n"
        "api.KernelProvider.initialize(0, _lib=lib)\n"
        "import cases\n"
        "def launches():\n"
        "    v, w = ctypes.c_uint64(), ctypes.c_uint64()\n"
        "    lib.troyhip_stat(b'ntt1_fp_launches', ctypes.byref(v)); lib.troyhip_stat(b'ntt1_int_launches', ctypes.byref(w))\n"
        "    return v.value + w.value\n"
        "if sys.argv[1] == 'ntt':\n"
        "  for logn in (12, 13, 14):\n"
        "    N = 1 << logn\n"
        "    kp = api.CoeffModulus.Create(N, [60, 50, 58, 40, 60])\n"
        "    ctx = api.SEALContext(api.BFV, N, kp, api.PlainModulus.Batching(N, 20))\n"
        "    primes = kp[:4] + [int(ctx.behz_bases(4)[0][0])]\n"   # + a 61-bit BEHZ prime (guarded butterflies)
        "    rows = 3 * len(primes) * 2\n"
        "    x = synth.uniform_rows(7, primes, rows, N, inner=2)\n"
        "    buf = api.DeviceBuffer.from_numpy(x)\n"
        "    l0 = launches()\n"
        "    ctx.ntt(buf, rows, primes, inner=2)\n"
        "    assert launches() >= l0 + 3\n"   # one launch per prime class
        "    y = buf.to_numpy().reshape(rows, N)\n"
        "    for r in range(rows):\n"
        "        assert np.array_equal(y[r], oracle.ntt_standalone(N, primes[(r // 2) %% len(primes)], x[r], 1)), (logn, r)\n"
        "    ctx.ntt(buf, rows, primes, inner=2, inverse=True)\n"
        "    assert np.array_equal(buf.to_numpy().reshape(rows, N), x)\n"
        "    rows2 = 2 * len(primes)\n"
        "    e = np.zeros((rows2, N), dtype=np.uint64)\n"
        "    for r in range(rows2):\n"
        "        p = primes[(r // 2) %% len(primes)]\n"
        "        e[r] = p - 1\n"
        "        if r %% 2: e[r, 1::2] = 0\n"
        "    buf2 = api.DeviceBuffer.from_numpy(e)\n"
        "    ctx.ntt(buf2, rows2, primes, inner=2, inverse=True)\n"
        "    z = buf2.to_numpy().reshape(rows2, N)\n"
        "    for r in range(rows2):\n"
        "        assert np.array_equal(z[r], oracle.ntt_standalone(N, primes[(r // 2) %% len(primes)], e[r], 3)), (logn, r)\n"
        "    ctx.ntt(buf2, rows2, primes, inner=2)\n"
        "    assert np.array_equal(buf2.to_numpy().reshape(rows2, N), e)\n"
        "  print('ok')\n"
        "else:\n"
        "  l0 = launches()\n"
        "  print(' '.join(cases.mul_relin_hash(n, batch=3) for n in ('cfgA_bfv_n4096_k3', 'cfgB_bfv_n8192_k5', 'bfv_n16384_k4')), launches() - l0)\n"
        % (ROOT, os.path.join(ROOT, 'tests'), EMUL))
    for rpw in ("2", "1"):
        env = dict(os.environ, TROYHIP_NTT1_RPW=rpw, TROYHIP_NTT="single", TROYHIP_AUX_BASE="reference")
        out = subprocess.run([sys.executable, str(script), "ntt"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    got = {}
    # xcd2 / xcd3: the XCD-aware workgroup order forced at this size, with the grouped list of the mod-down form (groups of 2 and of 3 primes: the
    # number of data primes is not a multiple of either, so the short last group runs too), one row per workgroup
    extra = {"single": {}, "twopass": {}, "xcd2": {"TROYHIP_NTT1_XCD": "1", "TROYHIP_NTT1_XCD_GROUP": "2", "TROYHIP_NTT1_RPW": "1"},
             "xcd3": {"TROYHIP_NTT1_XCD": "1", "TROYHIP_NTT1_XCD_GROUP": "3"}}
    for mode in ("single", "twopass", "xcd2", "xcd3"):
        env = dict(os.environ, TROYHIP_NTT="twopass" if mode == "twopass" else "single", **extra[mode])
        out = subprocess.run([sys.executable, str(script), "ops"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        got[mode] = out.stdout.split()[-4:]
    assert got["single"][:3] == got["twopass"][:3] == got["xcd2"][:3] == got["xcd3"][:3]
    assert int(got["single"][3]) >= 12 and int(got["twopass"][3]) == 0  # the single-pass kernels really ran (plain inverse, special limb, mod-down epilogue)
    assert int(got["xcd2"][3]) >= 12 and int(got["xcd3"][3]) >= 12
