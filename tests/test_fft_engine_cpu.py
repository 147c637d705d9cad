"""The compile-time mixed-radix butterfly engine (fft_amd/csrc/fft_regs_mixed.h) compiled for the host with g++ and
checked against numpy's FFT: forward, inverse, and forward->inverse through the output-position maps."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENGTHS = [2, 3, 4, 5, 7, 8, 6, 14, 10, 12, 15, 16, 20, 24, 25, 30, 32, 40, 48, 50, 56, 60, 64]


def _build(tmp_path_factory, name, extra=()):
    out = tmp_path_factory.mktemp(name) / "libfft_engine_host.so"
    cmd = ["g++", "-O1", "-std=c++17", "-shared", "-fPIC", *extra, "-I", os.path.join(ROOT, "tests", "host_shim"),
           "-I", os.path.join(ROOT, "fft_amd", "csrc"), os.path.join(ROOT, "tests", "host_shim", "fft_engine_host.cpp"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    lib = ctypes.CDLL(str(out))
    lib.fft_engine_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.fft_engine_run.restype = ctypes.c_int
    lib.fft_regs_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.fft_regs_run.restype = ctypes.c_int
    return lib


@pytest.fixture(scope="module")
def engine(tmp_path_factory):
    return _build(tmp_path_factory, "fft_engine")


@pytest.fixture(scope="module")
def scaled_engine(tmp_path_factory):
    """tools/fft_regs_mixed_scaled.h — the scaled-twiddle variant of the engine measured in round 3 (not shipped)."""
    header = os.path.join(ROOT, "tools", "fft_regs_mixed_scaled.h")
    return _build(tmp_path_factory, "fft_engine_scaled", [f'-DSFFT_ENGINE_HEADER="{header}"'])


def _run(lib, R, mode, x):
    buf = np.ascontiguousarray(x.astype(np.complex64)).view(np.float32).copy()
    assert lib.fft_engine_run(R, mode, buf.ctypes.data) == 0
    return buf.view(np.complex64)


@pytest.mark.parametrize("R", LENGTHS)
def test_forward_inverse_roundtrip(engine, R):
    rng = np.random.default_rng(R)
    x = rng.standard_normal(R) + 1j * rng.standard_normal(R)
    tol = 2e-6 * np.sqrt(R) * np.abs(np.fft.fft(x)).max()
    assert np.abs(_run(engine, R, 0, x) - np.fft.fft(x)).max() <= tol
    assert np.abs(_run(engine, R, 1, x) - np.fft.ifft(x) * R).max() <= tol
    assert np.abs(_run(engine, R, 2, x) - x * R).max() <= 4e-6 * R * np.abs(x).max()


def test_scaled_twiddle_experiment_computes_the_same_transforms(scaled_engine):
    for R in LENGTHS:
        rng = np.random.default_rng(1000 + R)
        x = rng.standard_normal(R) + 1j * rng.standard_normal(R)
        tol = 2e-6 * np.sqrt(R) * np.abs(np.fft.fft(x)).max()
        assert np.abs(_run(scaled_engine, R, 0, x) - np.fft.fft(x)).max() <= tol, R
        assert np.abs(_run(scaled_engine, R, 1, x) - np.fft.ifft(x) * R).max() <= tol, R
        assert np.abs(_run(scaled_engine, R, 2, x) - x * R).max() <= 4e-6 * R * np.abs(x).max(), R


@pytest.mark.parametrize("X", [16, 32, 64])
def test_two_factor_power_of_two_transforms_in_scaled_twiddle_form(engine, X):
    """fft_regs.h: fftA / fftB with the inter-stage twiddles applied to the second stage's inputs as unscaled rotations (2 FMAs) whose
    cos factors are folded into the butterfly's additions.  Against numpy, fp32 accuracy (a few 1e-7 of the spectrum's magnitude)."""
    rng = np.random.default_rng(100 + X)

    def run(mode, x):
        buf = np.ascontiguousarray(x.astype(np.complex64)).view(np.float32).copy()
        assert engine.fft_regs_run(X, mode, buf.ctypes.data) == 0
        return buf.view(np.complex64)

    for trial in range(8):
        x = rng.standard_normal(X) + 1j * rng.standard_normal(X)
        if trial == 0:
            x = np.zeros(X, complex); x[1] = 1.0                 # a pure twiddle pattern: every output is a root of unity
        F = np.fft.fft(x)
        tol = 1e-6 * max(np.abs(F).max(), 1.0)
        assert np.abs(run(0, x) - F).max() <= tol, "forward, type A"
        assert np.abs(run(1, x) - np.fft.ifft(x) * X).max() <= tol, "inverse, type A"
        assert np.abs(run(2, x) - x * X).max() <= 2e-6 * X * max(np.abs(x).max(), 1.0), "A forward -> B inverse"
        assert np.abs(run(3, x) - np.fft.ifft(x) * X).max() <= tol, "inverse, type B"
