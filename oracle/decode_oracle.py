"""CPU oracle for the prefill / decode path.  TEST INFRASTRUCTURE ONLY — see the header
of spectral_mix_oracle.py: nothing under fft_amd/ may import this.

Restates, statement by statement and in the same float32 evaluation order (the phases are formed from float32 products
of large arguments, so the order is part of the result):

  PrefixFFTCache.__init__ / prefill / decode_step      /root/reference/spectre.py:745-767, :769-783, :786-814
  SpectreHead.decode_step                              spectre.py:564-611
  interp_complex_1d (cubic branch)                     spectre.py:38-61
  ComplexModReLU.forward                               spectre.py:109-121
  pruned_irfft_single                                  spectre.py:614-655   (including the Nyquist term that is multiplied
                                                       by (-1)^pos a second time, :650 — a drop-in reproduces it)

Pinned by tests/test_decode_cpu.py against the fixtures tests/golden/g10_decode_*.npz (outputs of the reference itself).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class PrefixFFTCacheOracle:
    def __init__(self, n_fft: int, embed_dim: int):                       # spectre.py:745-767
        self.N, self.d = n_fft, embed_dim
        self.prefix_fft = torch.zeros(n_fft // 2 + 1, embed_dim, dtype=torch.cfloat)
        self.V_buf = torch.zeros(n_fft, embed_dim)
        self.Q_buf = torch.zeros_like(self.V_buf)
        self.sum_q = torch.zeros(embed_dim)
        self.t = -1
        self.freq_k = torch.arange(n_fft // 2 + 1, dtype=torch.float32)
        self.omega = -2 * math.pi / n_fft

    def prefill(self, Q, V):                                              # spectre.py:769-783
        L = V.size(0)
        V_pad = F.pad(V, (0, 0, 0, self.N - L))
        self.prefix_fft.copy_(torch.fft.rfft(V_pad, dim=0))
        self.V_buf[:L].copy_(V)
        self.Q_buf[:L].copy_(Q)
        self.sum_q = Q.sum(dim=0)
        self.t = L - 1

    def decode_step(self, q_t, v_t):                                      # spectre.py:786-814
        self.t += 1
        j = self.t % self.N
        v_old = self.V_buf[j]
        if self.t >= self.N:
            phase = torch.exp(1j * self.omega * self.freq_k * j)
            self.prefix_fft -= phase.unsqueeze(-1) * v_old
        phase_new = torch.exp(1j * self.omega * self.freq_k * self.t)
        self.prefix_fft += phase_new.unsqueeze(-1) * v_t
        self.V_buf[j] = v_t
        q_old = self.Q_buf[j]
        self.Q_buf[j] = q_t
        self.sum_q += q_t - (q_old if self.t >= self.N else 0.0)
        return self.prefix_fft, self.sum_q


def interp_cubic_oracle(x, size):                                         # spectre.py:38-61
    B, G, K = x.shape
    real_imag = torch.stack([x.real, x.imag], dim=1).reshape(B * G, 2, 1, K)
    grid_x = torch.linspace(-1, 1, size)
    grid = grid_x.view(1, 1, size, 1).expand(B * G, 1, size, 1)
    grid_2d = torch.cat([grid, torch.zeros_like(grid)], dim=-1)
    interp = F.grid_sample(real_imag, grid_2d, mode="bicubic", padding_mode="border", align_corners=True)
    return torch.complex(interp[:, 0, 0, :], interp[:, 1, 0, :]).view(B, G, size)


def modrelu_oracle(z, bias, eps):                                         # spectre.py:109-121
    mag = torch.abs(z)
    denom = torch.sqrt(mag.square() + eps.square())
    return z * (F.relu(mag + bias) / denom)


def pruned_irfft_single_oracle(X_half, n, pos):                           # spectre.py:614-655
    F_half, d = X_half.shape
    k = torch.arange(F_half, dtype=X_half.real.dtype)
    phase = 2 * math.pi * k * pos / n
    cos_phase, sin_phase = torch.cos(phase).unsqueeze(1), torch.sin(phase).unsqueeze(1)
    contrib = X_half.real * cos_phase - X_half.imag * sin_phase
    result = contrib[0]
    if n % 2 == 0:
        result = result + 2 * contrib[1:-1].sum(dim=0)
        result = result + contrib[-1] * ((-1) ** pos)
    else:
        result = result + 2 * contrib[1:].sum(dim=0)
    return result / n


@torch.no_grad()
def decode_step_oracle(head, q_t, v_t, cache: PrefixFFTCacheOracle):      # spectre.py:564-611
    """`head` supplies the learned pieces as plain torch modules / tensors: q_norm, gate_mlp, modrelu.bias, modrelu.eps,
    and the integers G, B (anchors per group), F_half, d_g."""
    prefix_fft, sum_q = cache.decode_step(q_t, v_t)
    descr = head.q_norm((sum_q / cache.N).unsqueeze(0)).squeeze(0)
    gate_anchor = torch.view_as_complex(head.gate_mlp(descr).view(head.G, head.B, 2))
    gate_half = interp_cubic_oracle(gate_anchor.unsqueeze(0), head.F_half).squeeze(0)
    gate_half = modrelu_oracle(gate_half.flatten(), head.modrelu.bias, head.modrelu.eps).view_as(gate_half)
    k = torch.arange(head.F_half)
    j = cache.t % cache.N
    phase = torch.exp(1j * 2 * math.pi * k * (cache.t - j) / cache.N)
    gate_half = gate_half * phase.unsqueeze(0)
    gate_broadcast = gate_half.permute(1, 0).repeat_interleave(head.d_g, dim=1)
    mixed_half = gate_broadcast * prefix_fft
    return pruned_irfft_single_oracle(mixed_half, cache.N, cache.t % cache.N)
