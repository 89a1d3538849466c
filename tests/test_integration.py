"""The reference-side binding (integration/pbwtGpu.c) compiled and linked for real.

CPU: it compiles warning-free against the reference's own pbwt.h (skipped where /root/reference is absent), both
stand-alone and as the unity build; the resulting library — the reference's objects + the binding + libpbwtgpu.so —
links, loads and exports the replaced entry points.
GPU: the reference's OWN pbwtLongMatches / matchSequencesDynamic / reportMatch / -check run with the device underneath
and reproduce the goldens the CPU reference wrote (tests/refgpu_checks.py, in a fresh process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIB = os.path.join(ROOT, "oracle", "_ref", "libpbwtref_gpu.so")
REPLACED = ["matchMaximalWithin", "matchSequencesSweep", "matchSequencesSweepSparse", "pbwtBuildFromBitColumns", "pbwtCursorAtSite"]


def nm(path, flags="-g"):
    out = subprocess.check_output(["nm", flags, path], text=True)
    return {ln.split()[-1]: ln.split()[-2] for ln in out.splitlines() if len(ln.split()) >= 2}


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference headers (build container only)")
def test_binding_compiles_against_reference_headers(tmp_path):
    inc = ["-I" + REF, "-I" + os.path.join(ROOT, "include")]
    o1 = str(tmp_path / "pbwtGpu.o")
    # the binding's own code is warning-free; the reference's headers are not ours to fix (hash.h's unused static)
    subprocess.check_call(["gcc", "-c", "-O2", "-fPIC", "-Wall", "-Werror", "-Wno-unused-variable", "-o", o1] + inc +
                          [os.path.join(ROOT, "integration", "pbwtGpu.c")])
    syms = nm(o1)
    for s in REPLACED + ["matchLongWithin2"]:
        assert syms.get(s) == "T", s
    o2 = str(tmp_path / "pbwtMatchGpu.o")
    subprocess.check_call(["gcc", "-c", "-O2", "-fPIC", "-w", "-o", o2] + inc + [os.path.join(ROOT, "integration", "pbwtMatchGpu.c")])
    syms = nm(o2)
    for s in REPLACED + ["pbwtLongMatches", "matchSequencesDynamic", "matchMaximalWithin_cpu"]:
        assert syms.get(s) == "T", s
    assert syms.get("pbwtamd_max_within") == "U" and syms.get("pbwtamd_cursor_at") == "U"
    # the reference's pbwtLongMatches in that object calls the replacement, not the renamed CPU body
    dis = subprocess.check_output(["objdump", "-dr", "--no-show-raw-insn", o2], text=True)
    body = dis.split("<pbwtLongMatches>:")[1].split("\n\n")[0]
    assert "matchMaximalWithin_cpu" not in body and "matchMaximalWithin" in body
    import oracle
    oracle.build()                                 # (re)links oracle/_ref/libpbwtref_gpu.so
    assert os.path.exists(LIB)


def test_gpu_bound_reference_library_loads():
    if not os.path.exists(LIB):
        pytest.skip("oracle/_ref/libpbwtref_gpu.so not built (needs /root/reference)")
    import ctypes as C
    import pbwt_amd
    pbwt_amd.load_library()
    L = C.CDLL(LIB)
    for s in REPLACED + ["pbwtLongMatches", "matchSequencesDynamic", "refgpu_match_dynamic_to_file"]:
        assert hasattr(L, s), s
    assert L.refgpu_is_gpu_build() == 1
    undefined = [k for k, v in nm(LIB, "-D").items() if v == "U" and k.startswith("pbwtamd_")]
    assert "pbwtamd_max_within" in undefined and "pbwtamd_match_sweep" in undefined


@pytest.mark.gpu
def test_reference_callers_run_on_the_gpu():
    assert os.path.exists(LIB), "oracle/_ref/libpbwtref_gpu.so did not travel with the repo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "refgpu_checks.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "REFGPU_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
