"""Size-independent identities of the hot path, checked on the oracle (CPU).  The same identities are
used on the GPU at BASELINE.json's full sizes, where the oracle itself would be too slow."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle.spectral_mix_oracle import assert_close, spectral_mix_dft64, spectral_mix_numpy


def _rand(rng, B, N, D, G, n_fft, zero_frac=0.2):
    V = rng.standard_normal((B, N, D)).astype(np.float32)
    F = n_fft // 2 + 1
    g = (rng.standard_normal((B, G, F)) + 1j * rng.standard_normal((B, G, F))) * 0.3
    g = g * (rng.random((B, G, F)) >= zero_frac)
    return V, g.astype(np.complex64)


@settings(max_examples=25, deadline=None)
@given(n_fft=st.integers(2, 96), extra=st.integers(-5, 5), G=st.sampled_from([1, 2, 3]), seed=st.integers(0, 2**16))
def test_fft_oracle_equals_dft_oracle(n_fft, extra, G, seed):
    rng = np.random.default_rng(seed)
    N = max(1, n_fft + extra)
    V, g = _rand(rng, 2, N, 2 * G, G, n_fft)
    mem = (rng.standard_normal((n_fft // 2 + 1, 2 * G)) + 1j * rng.standard_normal((n_fft // 2 + 1, 2 * G))).astype(np.complex64)
    a = spectral_mix_numpy(V, g, mem, n_fft)
    b = spectral_mix_dft64(V, g, mem, n_fft)
    assert a.shape == (2, min(N, n_fft), 2 * G)
    assert_close(a, b, rtol=1e-9, atol_rms=1e-9, what="numpy vs DFT64")


@pytest.mark.parametrize("n_fft", [16, 15, 60])
def test_unit_gate_is_identity(n_fft):
    rng = np.random.default_rng(1)
    V, _ = _rand(rng, 2, n_fft, 4, 2, n_fft)
    g = np.ones((2, 2, n_fft // 2 + 1), np.complex64)
    assert_close(spectral_mix_numpy(V, g, None, n_fft), V, rtol=1e-9, atol_rms=1e-9)


@pytest.mark.parametrize("n_fft", [32, 21])
def test_imag_of_dc_and_nyquist_is_ignored(n_fft):
    rng = np.random.default_rng(2)
    V, g = _rand(rng, 2, n_fft, 4, 2, n_fft, 0.0)
    g2 = g.copy()
    g2[..., 0] = g2[..., 0].real + 5j
    if n_fft % 2 == 0:
        g2[..., -1] = g2[..., -1].real - 3j
    assert_close(spectral_mix_numpy(V, g2, None, n_fft), spectral_mix_numpy(V, g, None, n_fft), rtol=1e-9, atol_rms=1e-9)


def test_linearity_and_memory_superposition():
    rng = np.random.default_rng(3)
    n = 48
    V1, g = _rand(rng, 2, n, 6, 3, n)
    V2, _ = _rand(rng, 2, n, 6, 3, n)
    V1, V2 = V1.astype(np.float64), V2.astype(np.float64)      # keep the linear combination exact
    mem = (rng.standard_normal((n // 2 + 1, 6)) + 1j * rng.standard_normal((n // 2 + 1, 6))).astype(np.complex64)
    y12 = spectral_mix_numpy(2.0 * V1 - 3.0 * V2, g, None, n)
    assert_close(y12, 2.0 * spectral_mix_numpy(V1, g, None, n) - 3.0 * spectral_mix_numpy(V2, g, None, n), rtol=1e-9, atol_rms=1e-9)
    # memory is an additive, batch-invariant term: mix(V, g, mem) = mix(V, g) + mix(0, g, mem)
    ym = spectral_mix_numpy(V1, g, mem, n)
    assert_close(ym, spectral_mix_numpy(V1, g, None, n) + spectral_mix_numpy(np.zeros_like(V1), g, mem, n), rtol=1e-9, atol_rms=1e-9)


def test_circular_convolution_identity_and_shift_equivariance():
    """out = circular convolution of V[b,:,c] with the REAL kernel irfft(gate[b, c//d_g]) (SURVEY §4 id. 2)."""
    rng = np.random.default_rng(4)
    n, D, G = 40, 4, 2
    V, g = _rand(rng, 2, n, D, G, n, 0.0)
    y = spectral_mix_numpy(V, g, None, n)
    gh = g.astype(np.complex128).copy()
    h = np.fft.irfft(gh, n=n, axis=-1)                       # (B, G, n) real impulse responses
    ref = np.zeros_like(y)
    for b in range(2):
        for c in range(D):
            k = h[b, c // (D // G)]
            for i in range(n):
                ref[b, i, c] = sum(k[j] * V[b, (i - j) % n, c] for j in range(n))
    assert_close(y, ref, rtol=1e-8, atol_rms=1e-8)
    s = 7
    assert_close(spectral_mix_numpy(np.roll(V, s, axis=1), g, None, n), np.roll(y, s, axis=1), rtol=1e-9, atol_rms=1e-9)


def test_parseval_on_unfiltered_path():
    rng = np.random.default_rng(5)
    n = 64
    V, _ = _rand(rng, 1, n, 2, 1, n)
    phase = np.exp(1j * rng.uniform(0, 2 * np.pi, (1, 1, n // 2 + 1))).astype(np.complex64)   # |gate| = 1
    phase[..., 0] = 1.0
    phase[..., -1] = 1.0
    y = spectral_mix_numpy(V, phase, None, n)
    assert np.allclose((y ** 2).sum(axis=1), (V.astype(np.float64) ** 2).sum(axis=1), rtol=1e-9)
