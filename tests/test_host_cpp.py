"""The C++ host mirror (include/kmer_index_amd/) — compiled everywhere, executed on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_host_api.bin")


def _compile():
    from kmer_index_amd import build
    build.build()
    libdir = os.path.join(ROOT, "kmer_index_amd")
    cmd = ["g++", "-std=c++20", "-O2", "-Wall", "-Wextra", "-Werror", f"-I{os.path.join(ROOT, 'include')}",
           os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp"), "-o", BIN, f"-L{libdir}", "-lkmx", f"-Wl,-rpath,{libdir}",
           "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return BIN


def test_host_mirror_compiles_against_the_c_abi():
    """Reference-shaped caller code (make_kmer_index<ks...>, search(q).to_vector()) compiles with g++ -std=c++20."""
    assert os.path.exists(_compile())


def test_host_flatten_under_sanitizers():
    """kmx_host.cpp (flatten of one element, tables, aligned copy) against a std::map of buckets, built with
    AddressSanitizer + UBSan and run on the CPU: the host half of kmx_index_build needs no GPU."""
    csrc = os.path.join(ROOT, "kmer_index_amd", "csrc")
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_flatten.bin")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall", "-Wextra",
           f"-I{csrc}", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "tests", "cpp", "test_host_flatten.cpp"),
           os.path.join(csrc, "kmx_host.cpp"), "-o", exe, "-pthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "host flatten ok" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]


@pytest.mark.gpu
def test_host_mirror_runs_on_gpu():
    exe = _compile()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host api ok" in res.stdout
