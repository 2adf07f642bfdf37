import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fisher-nerf-customized_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ref
    ref.build()
    return ref


@pytest.fixture(scope="session")
def harness():
    """g++ build of the kernels' host/device-neutral arithmetic (csrc/fr_math.h) for CPU-side checks."""
    import ctypes
    hdir = os.path.join(ROOT, "tests", "harness")
    so = os.path.join(hdir, "libfr_math_harness.so")
    srcs = [os.path.join(hdir, "fr_math_harness.cpp"), os.path.join(PKG, "csrc", "fr_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                               "-o", so, srcs[0]])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
