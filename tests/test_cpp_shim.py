"""include/troyn.hpp -- the C++ mirror of the reference's troyn:: interface (src/troy_cuda.cuh) -- compiled with plain g++
from user-style code (tests/cpp/test_troyn.cpp).  CPU: linked against the emulator build of the same sources
(host logic + ABI shape);  GPU: linked against libtroyhip.so and run on the device."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_troyn.cpp")
APP = os.path.join(ROOT, "tests", "cpp", "test_troyn_app.cpp")  # include/troyn_app.hpp: the flows of the reference's test/app/linear_ckks.cu


def _build(out, libdir, libfile, src=SRC):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           os.path.join(libdir, libfile), "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


LINEAR = os.path.join(ROOT, "tests", "cpp", "test_troyn_linear.cpp")  # include/troyn_linear.hpp: the flows of the reference's test/app/linear.cu (BFV)


def _run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAIL" not in r.stdout


def test_troyn_header_on_emulator(tmp_path):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    exe = str(tmp_path / "test_troyn_emul")
    _build(exe, os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so")
    _run(exe)


@pytest.mark.gpu
def test_troyn_header_on_gpu(tmp_path):
    exe = str(tmp_path / "test_troyn")
    _build(exe, os.path.join(ROOT, "troy_amd"), "libtroyhip.so")
    _run(exe)


def test_troyn_app_on_emulator(tmp_path):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    exe = str(tmp_path / "test_troyn_app_emul")
    _build(exe, os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so", APP)
    _run(exe)


@pytest.mark.gpu
def test_troyn_app_on_gpu(tmp_path):
    exe = str(tmp_path / "test_troyn_app")
    _build(exe, os.path.join(ROOT, "troy_amd"), "libtroyhip.so", APP)
    _run(exe)


def test_troyn_linear_on_emulator(tmp_path):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    exe = str(tmp_path / "test_troyn_linear_emul")
    _build(exe, os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so", LINEAR)
    _run(exe, "4096")


@pytest.mark.gpu
def test_troyn_linear_on_gpu(tmp_path):
    exe = str(tmp_path / "test_troyn_linear")
    _build(exe, os.path.join(ROOT, "troy_amd"), "libtroyhip.so", LINEAR)
    _run(exe, "16384")  # the degree of the reference's own run (test/app/linear.cu:578)


BATCH = os.path.join(ROOT, "tests", "cpp", "test_troyn_batch.cpp")  # the slab-batched Evaluator forms against the per-ciphertext forms, limb for limb
BENCH = os.path.join(ROOT, "tests", "cpp", "bench_troyn.cpp")       # multiply + relinearize ops/s through troyn.hpp (single / loop / batch), each verified


def test_troyn_batch_on_emulator(tmp_path):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    exe = str(tmp_path / "test_troyn_batch_emul")
    _build(exe, os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so", BATCH)
    _run(exe, "1024", "3")


@pytest.mark.gpu
def test_troyn_batch_on_gpu(tmp_path):
    exe = str(tmp_path / "test_troyn_batch")
    _build(exe, os.path.join(ROOT, "troy_amd"), "libtroyhip.so", BATCH)
    _run(exe, "8192", "7")
    _run(exe, "32768", "3")  # the single-pass transforms and their fused epilogues under a strided batch


def test_bench_troyn_on_emulator(tmp_path):
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    exe = str(tmp_path / "bench_troyn_emul")
    _build(exe, os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so", BENCH)
    _run(exe, "bfv_n4096_l2", "1", "1", "2")


@pytest.mark.gpu
def test_bench_troyn_on_gpu(tmp_path):
    exe = str(tmp_path / "bench_troyn")
    _build(exe, os.path.join(ROOT, "troy_amd"), "libtroyhip.so", BENCH)
    r = subprocess.run([exe, "bfv_n8192_l4", "3", "1", "8", "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL OK" in r.stdout and '"verified": false' not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


DEVICES = os.path.join(ROOT, "tests", "cpp", "test_troyn_devices.cpp")  # include/troyn_devices.hpp: a batch sharded over several contexts / devices / host threads


def _build_threads(out, libdir, libfile, src):
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-o", out,
           os.path.join(libdir, libfile), "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_troyn_devices_on_emulator(tmp_path):
    """KernelProvider's statics, DeviceGroup / shardBatch: three members on three VIRTUAL devices of the emulator build (per-device pools, contexts bound to
    their device, peer copies, a host thread per member) and three members sharing device 0 -- both equal the single-context batch limb for limb"""
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "troy_amd", "csrc"), "emul"])
    exe = str(tmp_path / "test_troyn_devices_emul")
    _build_threads(exe, os.path.join(ROOT, "tests", "emul"), "libtroyhip_emul.so", DEVICES)
    os.environ["HIP_EMUL_DEVICES"] = "3"
    try:
        _run(exe, "1024", "3", "1")
    finally:
        del os.environ["HIP_EMUL_DEVICES"]
    _run(exe, "1024", "3", "0")
    _run(exe, "1024", "1", "0")


@pytest.mark.gpu
def test_troyn_devices_on_gpu(tmp_path):
    """the same on the device: four members (contexts, scratch arenas, host threads) sharing GPU 0; on a node with more GPUs also one member per GPU"""
    exe = str(tmp_path / "test_troyn_devices")
    _build_threads(exe, os.path.join(ROOT, "troy_amd"), "libtroyhip.so", DEVICES)
    _run(exe, "8192", "4", "0")
    _run(exe, "32768", "2", "0")
    import torch
    n = torch.cuda.device_count()
    if n > 1:
        _run(exe, "8192", str(min(n, 8)), "1")


def test_fp64_arithmetic_and_bound_walk_on_cpu(tmp_path):
    """troy_amd/csrc/fpmod.h on the host: exact products within their stated magnitude, and a plain-loop model of the transforms under the masks of
    fp_plan / fp_plan_inv -- every intermediate value an exact integer below 2^53 and below the walk's bound, final residues equal to the integer
    transform's, unschedulable round shapes refused (tests/cpp/test_fp_plan.cpp)"""
    exe = str(tmp_path / "test_fp_plan")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-DTROYHIP_CPU_EMUL", "-I" + os.path.join(ROOT, "tests", "emul"), "-I" + os.path.join(ROOT, "troy_amd", "csrc"),
                    "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_fp_plan.cpp"), "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-3000:]
