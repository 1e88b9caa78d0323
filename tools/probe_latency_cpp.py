#!/usr/bin/env python3
"""Builds and runs tools/probe_latency.cpp (the C++ mirror's single-query latency).  KMX_NO_SMALL=1 shows the general pipeline."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmer_index_amd import build  # noqa: E402

build.build()
libdir = os.path.join(ROOT, "kmer_index_amd")
exe = os.path.join(ROOT, "tools", "micro", "probe_latency.bin")
os.makedirs(os.path.dirname(exe), exist_ok=True)
subprocess.check_call(["g++", "-std=c++20", "-O2", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "tools", "probe_latency.cpp"), "-o", exe,
                       f"-L{libdir}", "-lkmx", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
sys.exit(subprocess.call([exe]))
