"""Builds kmer_index_amd/libkmx.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkmx.so")
SOURCES = ["kmx_kernels.hip", "kmx_capi.hip", "kmx_build_sort.hip", "kmx_host.cpp"]
HEADERS = ["kmx_types.h", "kmx_host.h", "kmx_kernels.h", os.path.join("..", "..", "include", "kmx.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, fill_e: int | None = None, verbose: bool = False, checked: bool = False) -> str:
    """Compile every HIP source for gfx950.  hipcc cross-compiles, so no GPU is needed."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-Wall", "-Wno-unused-result", "-x", "hip"]
    if fill_e:
        cmd.append(f"-DKMX_FILL_E={fill_e}")
    for knob in ("KMX_PSORT_MULTIWAY_RUNS", "KMX_LOOKUP_OCC", "KMX_PSB_CPT", "KMX_PMERGE_MIN_AVG", "KMX_PMERGE_REG_RUNS", "KMX_PMERGE_REG_LEN", "KMX_PSORT_BAND_MIN", "KMX_SPLIT_TILE", "KMX_SPLIT_SCATTER_THREADS", "KMX_SPLIT", "KMX_NT_QUAD_LOADS", "KMX_PLAIN_QUAD_STORES", "KMX_VWAVE_OCC", "KMX_BAND", "KMX_MID_THREADS", "KMX_MID_OCC"):      # tuning experiments only
        if os.environ.get(knob):
            cmd.append(f"-D{knob}={int(os.environ[knob])}")
    if os.environ.get("KMX_PHASE_TIMING"):                                      # measurement build: tools/probe_phases.py
        cmd.append("-DKMX_PHASE_TIMING=1")
    if checked or os.environ.get("KMX_CHECKED"):
        cmd.append("-DKMX_CHECKED=1")
    cmd += [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    if verbose and (res.stdout or res.stderr):
        print(res.stdout + res.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
