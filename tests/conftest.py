import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box with -m gpu)")


# Tests that assert on TIMING (coarse guards against code-generation pathologies) run after everything else: under `-x` a timing flake on
# a noisy box must not hide parity tests that were collected behind it.  They also retry before they fail (see the modules).
TIMING_MODULES = ("test_perf_sanity_gpu.py", "test_bench_multirank_gpu.py")


def pytest_collection_modifyitems(config, items):
    late = [it for it in items if os.path.basename(str(it.fspath)) in TIMING_MODULES]
    if late:
        items[:] = [it for it in items if os.path.basename(str(it.fspath)) not in TIMING_MODULES] + late


def golden_files():
    """Forward / backward fixtures of the spectral mix (the decode and multi-head fixtures g10_*, g11_* have their own tests)."""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")) if not os.path.basename(f).startswith(("g10_decode", "g11_multihead", "g12_block", "g13_")))


def golden_ids():
    return [os.path.basename(f)[:-4] for f in golden_files()]


def load_golden(path):
    d = np.load(path)
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def built_library():
    """libspectre_hip.so, built if missing (hipcc cross-compiles without a GPU)."""
    from fft_amd import build
    return build.build()
