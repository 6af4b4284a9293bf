"""The reference-side binding of INTEGRATION.md (tests/cpp/ref_binding.hpp), compiled against the reference's own CPU headers
(/root/reference/src/troy_cpu.h) and linked with oracle/_ref/libtroyref.so (the reference CPU half built by oracle/Makefile) + the emulator
build of this library: reference KeyGenerator / Encryptor / encoders -> troyn::Evaluator (this repo's kernels) -> reference Decryptor, with
the result compared limb for limb with the reference's own troy::Evaluator on the same ciphertexts and keys.  Build container only: the
reference does not travel, so this is skipped wherever /root/reference or oracle/_ref is absent."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libtroyref.so")

needs_reference = pytest.mark.skipif(not (os.path.exists(os.path.join(REF_SRC, "troy_cpu.h")) and os.path.exists(REF_LIB)),
                                     reason="the reference sources / oracle/_ref are not present here")


@needs_reference
@pytest.mark.parametrize("n", [4096, 8192])
def test_reference_side_binding_on_emulator(tmp_path, n):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    emul = os.path.join(ROOT, "tests", "emul")
    exe = str(tmp_path / "test_ref_binding")
    cmd = ["g++", "-std=c++17", "-O2", "-w", "-I" + REF_SRC, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp"),
           os.path.join(ROOT, "tests", "cpp", "test_ref_binding.cpp"), "-o", exe, REF_LIB, os.path.join(emul, "libtroyhip_emul.so"),
           "-Wl,-rpath," + os.path.dirname(REF_LIB), "-Wl,-rpath," + emul]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
