import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_panels():
    return sorted(glob.glob(os.path.join(GOLDEN, "mosaic_*.npz")))


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_lib():
    """the product library; GPU tests fail (not skip) if it is missing or no device is usable"""
    import pbwt_amd
    pbwt_amd.load_library()
    return pbwt_amd


def parse_pbwt(path):
    raw = open(path, "rb").read()
    assert raw[:4] == b"PBW3"
    M, N = np.frombuffer(raw, "<i4", 2, 4)
    off = 12
    aFstart = np.frombuffer(raw, "<i4", M, off); off += 4 * M
    aFend = np.frombuffer(raw, "<i4", M, off); off += 4 * M
    nz = int(np.frombuffer(raw, "<i8", 1, off)[0]); off += 8 + 4
    yz = np.frombuffer(raw, np.uint8, nz, off)
    return int(M), int(N), aFstart.copy(), aFend.copy(), yz.copy()
