"""The reference's OWN GPU callers, unchanged, against include/ -- the drop-in claim tested on the reference's files, not on re-typed copies.

test/evaluator_cuda.cu (52 gtest cases), test/encryptor_cuda.cu (6), test/ckks_cuda.cu (2) -- the reference's only GPU acceptance tests
 -- plus the three mains test/timetest.cu, test/app/linear.cu, test/app/linear_ckks.cu are compiled FROM WHERE THEY LIE under
/root/reference by `make -C oracle dropin` (oracle/Makefile: each file goes through stdin with the working directory set to its twin under
include/dropin/, so its `#include "../src/troy_cuda.cuh"` finds include/troy_cuda.cuh; googletest is tests/cpp/gtest_shim).  Nothing of the
reference is copied into the repository; the built binaries live under oracle/_ref/dropin/ (git-ignored, travels to the GPU box like oracle/_ref).

CPU suite (build container only: needs /root/reference): build, then all 60 cases on the host emulator build of the library; the mains must link.
GPU suite: the same objects linked with libtroyhip.so -- the 60 cases and the three mains on the device.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "oracle", "_ref", "dropin")
CASES = 60  # 52 + 6 + 2 TEST() blocks in the three reference files

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "test", "evaluator_cuda.cu")), reason="reference sources only exist in the build container")


def _summary(stdout):
    m = re.search(r"\[==========\] (\d+) tests ran, (\d+) failed", stdout)
    assert m, stdout[-3000:]
    return int(m.group(1)), int(m.group(2))


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), "dropin"])
    return OUT


@needs_reference
def test_reference_sources_count_sixty_cases():
    n = 0
    for f in ("evaluator_cuda.cu", "encryptor_cuda.cu", "ckks_cuda.cu"):
        n += len(re.findall(r"^\s*TEST\(", open(os.path.join(REF, "test", f)).read(), re.M))
    assert n == CASES


@needs_reference
def test_reference_gpu_tests_pass_unchanged_on_emulator(built):
    r = subprocess.run([os.path.join(built, "troytest_emul")], capture_output=True, text=True, timeout=1500, cwd=built)
    ran, failed = _summary(r.stdout)
    assert (ran, failed, r.returncode) == (CASES, 0, 0), r.stdout[-4000:]


@needs_reference
def test_reference_mains_link_unchanged(built):
    # test/timetest.cu (N = 16384), test/app/linear.cu (N = 16384, 64 x 128 x 256) and linear_ckks.cu (56 x 56 convolution) take tens of minutes on
    # the fiber emulator: there they only have to compile and link against include/ and the library's exports; the GPU suite runs them
    for name in ("timetest", "linear", "linear_ckks"):
        for flavour in ("emul", "gpu"):
            assert os.access(os.path.join(built, "%s_%s" % (name, flavour)), os.X_OK)


@needs_reference
def test_default_modulus_tables_match_reference():
    # include/troyn_hestd.inc is DATA generated from the reference's tables: every number of src/utils/globals.cpp and hestdparams.h must be in it
    inc = open(os.path.join(ROOT, "include", "troyn_hestd.inc")).read().lower()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(REF, "src", "utils", "globals.cpp")).read(), flags=re.S)
    primes = re.findall(r"0x[0-9a-fA-F]+", src)
    assert len(primes) > 60
    for p in primes:
        assert p.lower() + "ull" in inc, p
    hs = open(os.path.join(REF, "src", "utils", "hestdparams.h")).read()
    for sec in (128, 192, 256):  # the classical-security tables SecurityLevel::tc128 / tc192 / tc256 select (src/modulus.cpp:14-51)
        body = re.search(r"seal_he_std_parms_%d_tc\(.*?\{(.*?)return 0;\s*\}" % sec, hs, re.S).group(1)
        bits = re.findall(r"size_t\(\d+\):\s*return (\d+);", body)
        assert len(bits) == 6
        assert "{" + ", ".join(bits) + "}" in inc, (sec, bits)


@needs_reference
def test_reference_targets_build_with_cmake_against_the_package(tmp_path):
    """what a maintainer of the reference does instead of this suite's stdin recipe: find_package(troyhip) (cmake/troyhipConfig.cmake: imported target
    troyhip::troyhip = the library + include/) and the overlay project cmake/reference_overlay, which builds the reference's own test/ and test/app/ GPU
    targets -- troytest (the three *_cuda.cu files), timetest, linear, linear_ckks -- unchanged, as LANGUAGE CXX, from a mirror of symlinks in the BUILD tree.
    Here against the host emulator build of the library; the encryptor suite of the cmake-built troytest runs, the other binaries must link."""
    import shutil
    if not shutil.which("cmake"):
        pytest.skip("cmake is not installed")
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    build = str(tmp_path / "build")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    cfg = subprocess.run(["cmake", "-S", os.path.join(ROOT, "cmake", "reference_overlay"), "-B", build, *gen, "-Dtroyhip_DIR=" + os.path.join(ROOT, "cmake"),
                          "-DTROY_REFERENCE_DIR=" + REF, "-DTROYHIP_LIBRARY=" + os.path.join(ROOT, "tests", "emul", "libtroyhip_emul.so"), "-DCMAKE_BUILD_TYPE=Release"],
                         capture_output=True, text=True, timeout=600)
    assert cfg.returncode == 0, cfg.stdout[-2000:] + cfg.stderr[-2000:]
    bld = subprocess.run(["cmake", "--build", build, "-j8"], capture_output=True, text=True, timeout=1500)
    assert bld.returncode == 0, bld.stdout[-3000:] + bld.stderr[-3000:]
    for name in ("troytest", "timetest", "linear", "linear_ckks"):
        assert os.access(os.path.join(build, name), os.X_OK), name
    assert os.path.islink(os.path.join(build, "mirror", "test", "evaluator_cuda.cu"))  # the reference's file itself, not a copy
    r = subprocess.run([os.path.join(build, "troytest"), "--gtest_filter=EncryptorCudaTest"], capture_output=True, text=True, timeout=900)
    ran, failed = _summary(r.stdout)
    assert (ran, failed, r.returncode) == (6, 0, 0), r.stdout[-3000:]


def _gpu_binary(name):
    path = os.path.join(OUT, name + "_gpu")
    if not os.access(path, os.X_OK):
        pytest.skip("oracle/_ref/dropin was not prebuilt (build() does it where /root/reference exists)")
    return path


@pytest.mark.gpu
def test_reference_gpu_tests_pass_unchanged_on_gpu():
    r = subprocess.run([_gpu_binary("troytest")], capture_output=True, text=True, timeout=1500, cwd=OUT)
    ran, failed = _summary(r.stdout)
    assert (ran, failed, r.returncode) == (CASES, 0, 0), r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_reference_timetest_runs_unchanged_on_gpu():
    r = subprocess.run([_gpu_binary("timetest")], capture_output=True, text=True, timeout=1500, cwd=OUT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    for label in ("Encrypt", "Decrypt", "Multiply-inplace", "Relinearize-inplace", "Square-assign", "RotateRows-inplace", "MultiplyPlain-assign"):  # timer labels of its BFV run
        assert re.search(label, r.stdout, re.I), r.stdout[-3000:]


@pytest.mark.gpu
def test_reference_linear_apps_run_unchanged_on_gpu():
    # test/app/linear.cu: BFV 64 x 128 x 256 matmul with ciphertext weights through app/LinearHelper.cuh's interface -- exact arithmetic modulo t
    r = subprocess.run([_gpu_binary("linear")], capture_output=True, text=True, timeout=1500, cwd=OUT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    diffs = re.findall(r"Difference = (\S+)", r.stdout)
    assert diffs and all(float(d) == 0 for d in diffs), r.stdout[-3000:]
    # test/app/linear_ckks.cu: CKKS 1 x 256 x 64 x 56 x 56 convolution of reals in [0, 10) at scale 2^15 -- approximate.  Each of the 256 products per
    # output carries the two encoding roundings, (|x| + |w|) 2^-16 <= 30 * 1.5e-5, so the sum is within 256 * 4.6e-4 = 0.12 of the exact value at worst
    # (a random walk of them: ~0.03, which is what the run prints); the reference itself only prints the number
    r = subprocess.run([_gpu_binary("linear_ckks")], capture_output=True, text=True, timeout=1500, cwd=OUT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    diffs = re.findall(r"Difference = (\S+)", r.stdout)
    assert diffs and all(float(d) < 0.12 for d in diffs), r.stdout[-3000:]
