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


@pytest.mark.gpu
def test_host_mirror_runs_on_gpu():
    exe = _compile()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "host api ok" in res.stdout
