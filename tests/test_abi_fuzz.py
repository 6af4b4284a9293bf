"""C-ABI metadata fuzz (tools/abi_fuzz.py) on the CPU emulator build of the kernel sources: thousands of random, mostly invalid ciphertext descriptors
through the evaluator entry points -- every call returns TROYHIP_OK or an error code with a message, what is accepted was a plausible operand, nothing
faults.  (Round 3: this is how the division by zero of a CKKS add / sub with unequal correction factors was found.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 3, 12])
def test_abi_metadata_fuzz_on_emulator(seed):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    env = dict(os.environ, TROYHIP_LIB=os.path.join(ROOT, "tests", "emul", "libtroyhip_emul.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_fuzz.py"), "3000", str(seed)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "no fault" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
