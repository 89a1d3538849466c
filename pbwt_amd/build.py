"""compile pbwt_amd/csrc -> pbwt_amd/libpbwtgpu.so with hipcc for gfx950 (cross-compiles without a GPU)"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/pbwt_engine.hip"]
DEPS = ["csrc/pbwt_engine.hip", "csrc/pbwt_kernels.h", "../include/pbwt_amd.h"]
OUT = os.path.join(_HERE, "libpbwtgpu.so")


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(_HERE, d)) > t for d in DEPS)


def build_library(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", OUT] + [os.path.join(_HERE, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build_library(force=True, verbose=True)
