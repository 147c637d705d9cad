"""CPU oracle for the wavelet refinement of the multi-head layer.  TEST INFRASTRUCTURE ONLY — see the header of
spectral_mix_oracle.py: nothing under fft_amd/ may import this.

Restates in numpy (float64 by default), level by level:

  HaarDWT.forward, one level            /root/reference/spectre.py:190-219   circular left pad by one (:204), correlation with h0 = (s, s) and
                                                                            h1 = (-s, s), stride 2 (:208-209), s = 1 / sqrt 2 (:186-188)
  dwt_decompose                         spectre.py:288-312                   int(log2 L) levels on the running approximation, detail bands kept
  HaarIDWT.forward, one level           spectre.py:246-272                   transposed convolution with g0 = (s, s), g1 = (s, -s), stride 2
  dwt_reconstruct                       spectre.py:315-328                   from the coarsest level up
  WaveletRefinement.forward             spectre.py:834-887                   v + (R(v) * gate[:, None, :]) * on_mask

Because the analysis pads on the left, it pairs (x[2j-1], x[2j]) while the synthesis emits (y[2j], y[2j+1]): the round trip R is a fixed linear
operator that is NOT the identity (one level: 0..7 -> 0,7,2,1,4,3,6,5).  Sequence lengths that are not powers of two make the reference raise
(:271) as soon as a level has odd length; `haar_round_trip` raises ValueError for them.

Pinned by tests/test_wavelet_cpu.py against the fixtures tests/golden/g13_wavelet_*.npz (outputs of the reference's own WaveletRefinement).
"""
from __future__ import annotations

import math

import numpy as np

_S = 1.0 / math.sqrt(2.0)


def _analysis(x: np.ndarray):
    """One level along the last axis (spectre.py:204-209): returns (lo, hi), each half as long."""
    xp = np.concatenate((x[..., -1:], x), axis=-1)                 # F.pad(x, (1, 0), mode='circular')
    a, b = xp[..., 0:-1:2], xp[..., 1::2]                          # windows (xp[2j], xp[2j+1]) = (x[2j-1], x[2j])
    return (a + b) * _S, (b - a) * _S                              # h0 = (s, s); h1 = (-s, s)


def _synthesis(lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
    """One level (spectre.py:260-270): y[2j] = s lo[j] + s hi[j], y[2j+1] = s lo[j] - s hi[j]."""
    y = np.empty(lo.shape[:-1] + (2 * lo.shape[-1],), dtype=lo.dtype)
    y[..., 0::2] = (lo + hi) * _S
    y[..., 1::2] = (lo - hi) * _S
    return y


def haar_round_trip(x: np.ndarray) -> np.ndarray:
    """dwt_reconstruct(dwt_decompose(x)) along the last axis (spectre.py:288-328)."""
    L = x.shape[-1]
    if L < 1 or L & (L - 1):
        raise ValueError(f"sequence length {L}: the reference's Haar pair only works for powers of two")
    details = []
    for _ in range(int(math.log2(L))):                             # :296, :299
        x, hi = _analysis(x)
        details.append(hi)
        if x.shape[-1] <= 1:                                       # :307
            break
    for hi in reversed(details):                                   # :321-326
        x = _synthesis(x, hi)
    return x


def wavelet_refinement_numpy(v: np.ndarray, gate: np.ndarray, on_mask: np.ndarray, dtype=np.float64) -> np.ndarray:
    """v (B, N, d), gate (B, d) = gate_mlp(q_pool), on_mask (B,) bool -> v + (R(v) * gate) * on_mask (spectre.py:853-886)."""
    v = np.asarray(v, dtype=dtype)
    out = v.copy()
    for b in np.nonzero(np.asarray(on_mask).reshape(-1))[0]:
        r = haar_round_trip(v[b].T).T                              # the reference transposes to (d, N) and back (:858, :867)
        out[b] = v[b] + r * np.asarray(gate[b], dtype=dtype)[None, :]
    return out


def wavelet_gate_grad_numpy(v: np.ndarray, dout: np.ndarray, on_mask: np.ndarray) -> np.ndarray:
    """d/d(gate) (B, d) of sum(out * dout): the round trip is detached (:884), so it is on_mask * sum_n dout * R(v)."""
    v, dout = np.asarray(v, dtype=np.float64), np.asarray(dout, dtype=np.float64)
    g = np.zeros((v.shape[0], v.shape[2]))
    for b in np.nonzero(np.asarray(on_mask).reshape(-1))[0]:
        g[b] = (haar_round_trip(v[b].T).T * dout[b]).sum(axis=0)
    return g
