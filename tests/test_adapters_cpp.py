"""The C++ host side above the C-ABI (include/orbgpu_adapters.hpp) compiles with plain g++, links against
liborbgpu.so, and fails loudly without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_smoke")


def _build():
    lib_dir = os.path.join(ROOT, "multi_orbslam3_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"),
           "-o", EXE, "-L", lib_dir, "-lorbgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_adapters_compile_link_and_fail_loudly_without_gpu():
    _build()
    from multi_orbslam3_amd import _capi
    if _capi.load().orbg_device_count() > 0:
        pytest.skip("a GPU is present; see the gpu-marked test")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no usable HIP device" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_adapters_run_on_gpu():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
