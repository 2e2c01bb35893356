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


@pytest.mark.gpu
def test_one_process_per_gpu_program_with_the_one_rank_this_box_allows():
    """tests/cpp/rccl_ranks.cpp: the C ABI's one-process-per-GPU form as a stand-alone program (fork per rank, the communicator id through pipes,
    fzb_shard_ranges / fzb_corpus_upload / fzb_shard_comm_create / fzb_match_list_parallel_rccl, every receiver checks against fzb_match_list over
    the whole list).  `rccl_ranks N` needs N GPUs; here N = 1."""
    exe = os.path.join(ROOT, "tests", "cpp", "rccl_ranks")
    src = exe + ".cpp"
    lib = os.path.join(ROOT, "frizbee_amd", "libfrizbee_hip.so")
    if not os.path.exists(exe) or any(os.path.getmtime(f) > os.path.getmtime(exe) for f in (src, lib, os.path.join(ROOT, "include", "frizbee_hip.h"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", os.path.join(ROOT, "frizbee_amd"), "-lfrizbee_hip",
                               "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "frizbee_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, "1", "60000"], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl_ranks 1: ok" in r.stdout and r.stdout.count("equal to the single-GPU list") == 2, (r.stdout[-1500:], r.stderr[-3000:])

