"""Prefill / single-token decode for one head.

`PrefixFFTCache` mirrors /root/reference/spectre.py:731-814 (constructor, attribute names, `prefill`, `decode_step`);
`head_decode_step` is what `SpectreHead.decode_step` (spectre.py:564-611) runs.  The spectrum work goes through the C ABI:
`spectre_rfft_fwd` (prefill) and `spectre_decode_step` (ring update fused with the filter multiply and the one-row
inverse transform, `pruned_irfft_single`, spectre.py:614-655).  The d-element ring-buffer copies, the running query sum,
LayerNorm and the gate MLP stay PyTorch ops; the gate tail is `spectre_gate_fwd`.  HIP devices only.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch

from . import _native
from .functional import _DT, spectral_gate_fused


def rfft_prefill(V: torch.Tensor, n_fft: int) -> torch.Tensor:
    """(B, N, D) or (N, D) real -> (B, F, D) or (F, D) complex64: rfft along the sequence, zero-padded / truncated to n_fft."""
    lib = _native.load()
    if not V.is_cuda:
        raise RuntimeError("rfft_prefill runs on a HIP device only (no CPU path)")
    if V.dtype not in _DT:
        raise TypeError(f"V dtype {V.dtype} unsupported (float32 or bfloat16)")
    squeeze = V.dim() == 2
    if squeeze:
        V = V.unsqueeze(0)
    if V.dim() != 3:
        raise ValueError(f"V must be (B, N, D) or (N, D), got {tuple(V.shape)}")
    if V.stride(2) != 1:
        V = V.contiguous()
    B, N, D = V.shape
    spec = torch.empty(B, n_fft // 2 + 1, D, dtype=torch.complex64, device=V.device)
    a = _native.SpectreRfftArgs()
    a.v, a.spec = V.data_ptr(), spec.data_ptr()
    a.B, a.N_in, a.n_fft, a.D = B, N, n_fft, D
    a.v_sb, a.v_sn = V.stride(0), V.stride(1)
    a.in_dtype = _DT[V.dtype]
    a.device = V.device.index if V.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(V.device).cuda_stream
    _native.check(lib.spectre_rfft_fwd(ctypes.byref(a)), "spectre_rfft_fwd")
    return spec[0] if squeeze else spec


class PrefixFFTCache:
    """Sliding-window frequency cache for autoregressive decoding — same surface as spectre.py:731-814."""

    def __init__(self, n_fft: int, embed_dim: int, device=None):
        self.N = n_fft
        self.d = embed_dim
        if device is None:                                   # spectre.py:750-752
            raise ValueError("PrefixFFTCache requires an explicit device parameter. "
                             "Pass device=tensor.device from your input tensors.")
        self.device = torch.device(device)
        self.prefix_fft = torch.zeros(n_fft // 2 + 1, embed_dim, dtype=torch.cfloat, device=device)
        self.V_buf = torch.zeros(n_fft, embed_dim, device=device)
        self.Q_buf = torch.zeros_like(self.V_buf)
        self.sum_q = torch.zeros(embed_dim, device=device)
        self.t = -1                                          # last filled position
        self.freq_k = torch.arange(n_fft // 2 + 1, device=device, dtype=torch.float32)
        self.omega = -2 * math.pi / n_fft
        self._ws: Optional[torch.Tensor] = None

    def _require_hip(self):
        if self.device.type != "cuda":
            raise RuntimeError("fft_amd.PrefixFFTCache computes on a HIP device only (no CPU path)")

    def prefill(self, Q: torch.Tensor, V: torch.Tensor):
        """Initialise from a prompt: Q, V (L, d), L <= N (spectre.py:769-783)."""
        self._require_hip()
        L = V.size(0)
        if L > self.N:
            raise ValueError(f"prompt length {L} exceeds n_fft {self.N}")   # the reference's F.pad fails on this too
        self.prefix_fft.copy_(rfft_prefill(V.float(), self.N))
        self.V_buf[:L].copy_(V)
        self.Q_buf[:L].copy_(Q)
        self.sum_q = Q.sum(dim=0).to(torch.float32).contiguous()      # fp32 always: the fused step reads and writes it as d floats
        self.t = L - 1

    # -- one step: spectrum update (+ optional fused filter and one-row inverse), ring buffers, query sum -------------
    def _advance(self, q_t: torch.Tensor, v_t: torch.Tensor, gate_fn=None) -> Optional[torch.Tensor]:
        self._require_hip()
        lib = _native.load()
        t = self.t + 1
        j = t % self.N
        evict = t >= self.N
        v_t = v_t.to(torch.float32).contiguous()
        # Running query sum (spectre.py:809-813).  In the reference `q_old = self.Q_buf[j]` is a VIEW that is read only
        # after `self.Q_buf[j] = q_t` has overwritten it, so once the ring wraps the update is `q_t - q_t`: the sum stops
        # moving.  A drop-in has to produce the same descriptor, so the same value is formed here.
        # The gate is built from the updated sum (spectre.py:575-578), which does not depend on the spectrum.
        sum_q = self.sum_q + ((q_t - q_t) if evict else (q_t - 0.0))
        gate = gate_fn(sum_q, t, j) if gate_fn is not None else None
        out = None
        a = _native.SpectreDecodeArgs()
        a.prefix = self.prefix_fft.data_ptr()
        a.v_old = self.V_buf[j].data_ptr()                   # read by the kernel before the ring copy below (same stream)
        a.v_new = v_t.data_ptr()
        a.n_fft, a.d, a.t = self.N, self.d, t
        if gate is not None:
            gate = gate.contiguous()
            out = torch.empty(self.d, dtype=torch.float32, device=self.device)
            if self._ws is None:
                self._ws = torch.empty(lib.spectre_decode_workspace_bytes(self.N, self.d), dtype=torch.uint8, device=self.device)
            a.gate, a.out, a.workspace, a.G = gate.data_ptr(), out.data_ptr(), self._ws.data_ptr(), gate.shape[0]
        a.device = self.device.index if self.device.index is not None else torch.cuda.current_device()
        a.stream = torch.cuda.current_stream(self.device).cuda_stream
        _native.check(lib.spectre_decode_step(ctypes.byref(a)), "spectre_decode_step")
        self.V_buf[j] = v_t
        self.Q_buf[j] = q_t
        self.sum_q = sum_q
        self.t = t
        return out

    def decode_step(self, q_t: torch.Tensor, v_t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """State update only (spectre.py:786-814): returns (updated prefix_fft, updated sum_q)."""
        self._advance(q_t, v_t)
        return self.prefix_fft, self.sum_q


def _same_device(a, b) -> bool:
    a, b = torch.device(a), torch.device(b)
    if a.type != b.type:
        return False
    cur = torch.cuda.current_device() if a.type == "cuda" else 0
    return (a.index if a.index is not None else cur) == (b.index if b.index is not None else cur)


def _fused_head_ok(head) -> bool:
    """The single-call path needs the reference's stock gate MLP (Linear -> GELU(erf) -> Linear) and LayerNorm in fp32."""
    import torch.nn as nn
    mlp = head.gate_mlp
    return (isinstance(mlp, nn.Sequential) and len(mlp) == 3 and isinstance(mlp[0], nn.Linear) and isinstance(mlp[2], nn.Linear)
            and isinstance(mlp[1], nn.GELU) and getattr(mlp[1], "approximate", "none") == "none"
            and isinstance(head.q_norm, nn.LayerNorm) and head.q_norm.elementwise_affine and head.q_norm.bias is not None
            and all(p.dtype == torch.float32 and p.is_contiguous() for p in list(mlp.parameters()) + list(head.q_norm.parameters())))


@torch.no_grad()
def head_decode_step(head, q_t: torch.Tensor, v_t: torch.Tensor, cache: PrefixFFTCache) -> torch.Tensor:
    """SpectreHead.decode_step (spectre.py:564-611): mixed vector (d,) for the current time step, batch size 1.

    One C-ABI call (`spectre_decode_head_step`, four launches): running query sum -> LayerNorm -> gate MLP; cubic resample ->
    modReLU -> decode phase; spectrum update + filter + one-row inverse; partial sums + ring-buffer writes."""
    cache._require_hip()
    if head.use_toeplitz or not _fused_head_ok(head):            # (the single-call path has no Toeplitz step: anchors through PyTorch ops then)
        return _head_decode_step_ops(head, q_t, v_t, cache)
    # The C entry point takes raw pointers and reads every operand as d (or n_fft-derived) float32 values: refuse anything
    # whose size, dtype or device does not match, where the reference would raise a shape error.
    if cache.N != head.n_fft or cache.d != head.d:
        raise ValueError(f"cache built for (n_fft={cache.N}, d={cache.d}) but the head has (n_fft={head.n_fft}, d={head.d})")
    for name, x in (("q_t", q_t), ("v_t", v_t)):
        if x.numel() != cache.d or not _same_device(x.device, cache.device):
            raise ValueError(f"{name} must hold d={cache.d} values on {cache.device}, got {tuple(x.shape)} on {x.device}")
    bias = head.modrelu.bias
    if bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != head.G * head.F_half or not _same_device(bias.device, cache.device):
        return _head_decode_step_ops(head, q_t, v_t, cache)
    for buf, shape in ((cache.prefix_fft, (head.F_half, cache.d)), (cache.V_buf, (cache.N, cache.d)), (cache.Q_buf, (cache.N, cache.d))):
        if tuple(buf.shape) != shape or not buf.is_contiguous() or not _same_device(buf.device, cache.device):
            raise ValueError("PrefixFFTCache buffers were replaced by tensors of another shape, layout or device")
    if cache.V_buf.dtype != torch.float32 or cache.Q_buf.dtype != torch.float32 or cache.prefix_fft.dtype != torch.complex64:
        raise TypeError("PrefixFFTCache buffers must stay float32 / complex64")
    lib = _native.load()
    t = cache.t + 1
    q_t = q_t.reshape(-1).to(torch.float32).contiguous()
    v_t = v_t.reshape(-1).to(torch.float32).contiguous()
    if cache.sum_q.dtype != torch.float32 or cache.sum_q.numel() != cache.d or not _same_device(cache.sum_q.device, cache.device):
        cache.sum_q = cache.sum_q.reshape(-1).to(device=cache.device, dtype=torch.float32)
    out = torch.empty(cache.d, dtype=torch.float32, device=cache.device)
    need = lib.spectre_decode_head_workspace_bytes(cache.N, cache.d, head.G, head.B)
    if cache._ws is None or cache._ws.numel() < need:
        cache._ws = torch.empty(need, dtype=torch.uint8, device=cache.device)
    if not cache.sum_q.is_contiguous():
        cache.sum_q = cache.sum_q.contiguous()
    a = _native.SpectreDecodeHeadArgs()
    a.prefix, a.V_buf, a.Q_buf, a.sum_q = cache.prefix_fft.data_ptr(), cache.V_buf.data_ptr(), cache.Q_buf.data_ptr(), cache.sum_q.data_ptr()
    a.q_t, a.v_t, a.out, a.workspace = q_t.data_ptr(), v_t.data_ptr(), out.data_ptr(), cache._ws.data_ptr()
    a.ln_w, a.ln_b = head.q_norm.weight.data_ptr(), head.q_norm.bias.data_ptr()
    a.w1, a.b1 = head.gate_mlp[0].weight.data_ptr(), head.gate_mlp[0].bias.data_ptr()
    a.w2, a.b2 = head.gate_mlp[2].weight.data_ptr(), head.gate_mlp[2].bias.data_ptr()
    a.modrelu_bias = head.modrelu.bias.data_ptr()
    a.ln_eps, a.modrelu_eps = float(head.q_norm.eps), float(head.modrelu.eps_value)
    a.n_fft, a.d, a.G, a.K, a.h1, a.t = cache.N, cache.d, head.G, head.B, head.gate_mlp[0].out_features, t
    a.device = cache.device.index if cache.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(cache.device).cuda_stream
    _native.check(lib.spectre_decode_head_step(ctypes.byref(a)), "spectre_decode_head_step")
    cache.t = t
    return out


@torch.no_grad()
def _head_decode_step_ops(head, q_t: torch.Tensor, v_t: torch.Tensor, cache: PrefixFFTCache) -> torch.Tensor:
    """Same step with the descriptor / MLP as PyTorch ops (non-standard gate MLPs); spectrum work still on the HIP kernels."""

    def gate_fn(sum_q, t, j):
        descr = head.q_norm((sum_q / cache.N).unsqueeze(0)).squeeze(0)                       # :578
        anchors = torch.view_as_complex(head.gate_mlp(descr).view(1, head.G, head.B, 2))     # :579-580
        if head.use_toeplitz:                                                                 # :582-584
            from .spectre import complex_conv1d
            anchors = anchors + complex_conv1d(anchors, head.toeplitz_kernel, head.toeplitz_bw)
        phase = None
        if t != j:   # exp(1j*2*pi*k*(t-j)/N) is exactly 1 while t < N; afterwards the reference's float32 value is used
            k = torch.arange(head.F_half, device=anchors.device)
            phase = torch.exp(1j * 2 * math.pi * k * (t - j) / cache.N)                      # :593-596, same expression
        return spectral_gate_fused(anchors, head.modrelu.bias.detach(), head.modrelu.eps_value, head.F_half, phase)[0]

    return cache._advance(q_t, v_t, gate_fn)
