"""compile pbwt_amd/csrc -> pbwt_amd/libpbwtgpu.so with hipcc for gfx950 (cross-compiles without a GPU)"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/pbwt_engine.hip"]
DEPS = ["csrc/pbwt_engine.hip", "../include/pbwt_amd.h"] + sorted("csrc/" + f for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith((".h", ".inc")))
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
    if os.environ.get("PBWTAMD_MEASURE_BUILD"):            # measurement build: compiles in the result-corrupting probe switches
        cmd.insert(1, "-DPBWTAMD_MEASURE")
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


CLI_OUT = os.path.join(_HERE, "pbwt")
CLI_SRC = ["host/pbwt_cli.c", "host/pbwt_host.c"]


def build_cli(force=False, verbose=False):
    """the host C side: `pbwt_amd/pbwt`, the reference's CLI grammar over the C ABI (gcc, plain C)"""
    build_library(force=False)
    deps = CLI_SRC + ["host/pbwt_host.h", "../include/pbwt_amd.h"]
    if not force and os.path.exists(CLI_OUT) and all(os.path.getmtime(os.path.join(_HERE, d)) <= os.path.getmtime(CLI_OUT) for d in deps) \
            and os.path.getmtime(OUT) <= os.path.getmtime(CLI_OUT):
        return CLI_OUT
    cmd = ["gcc", "-O2", "-std=gnu11", "-Wall", "-o", CLI_OUT] + [os.path.join(_HERE, s) for s in CLI_SRC] + \
          ["-L" + _HERE, "-lpbwtgpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CLI_OUT


if __name__ == "__main__":
    build_cli(force=True, verbose=True)
    build_library(force=True, verbose=True)
