"""The C++ host side (include/frizbee_hip.hpp, the mirror of the reference's Rust API above the C ABI) compiled with g++ and
run: the host-only part here (defaults, query parser, panics with the reference's text, loud failure without a GPU), the
reference's own matcher tests on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_facade")


def build():
    src = os.path.join(ROOT, "tests", "cpp", "test_facade.cpp")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("frizbee_hip.hpp", "frizbee_hip.h")]
    lib = os.path.join(ROOT, "frizbee_amd", "libfrizbee_hip.so")
    if not os.path.exists(EXE) or any(os.path.getmtime(f) > os.path.getmtime(EXE) for f in [src, lib] + hdrs):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", EXE, "-L", os.path.join(ROOT, "frizbee_amd"),
                               "-lfrizbee_hip", "-Wl,-rpath," + os.path.join(ROOT, "frizbee_amd")])
    return EXE


def test_host_side_of_the_cpp_facade():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the gpu-marked test runs the whole program")
    r = subprocess.run([build()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_reference_matcher_tests_through_the_cpp_facade():
    r = subprocess.run([build(), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
