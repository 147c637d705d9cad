"""SpectreHead — drop-in for the reference layer with the spectral mix running as one HIP kernel.

Mirrors the `nn.Module` surface of `SpectreHead` in `/root/reference/spectre.py:400-557` (constructor
keywords, attribute / parameter / buffer names, `forward` signature and return convention), so a reference
`state_dict()` loads unchanged and callers (`SpectreMultiHead.forward`, spectre.py:712) need no edits.

What runs where
---------------
* projections `W_q`, `W_v` (spectre.py:502-503), pooling, LayerNorm and the gate MLP (spectre.py:511-516): stock
  PyTorch-ROCm ops (GEMMs and O(B*d) work).  The tail of the gate producer — cubic resample -> modReLU ->
  positional phase (spectre.py:518-536) — is one HIP launch in inference (`spectral_gate_fused`, SURVEY.md
  section 8(f) row N2) and the same PyTorch ops as the reference under autograd.  Exposed as `spectral_gate()`.
* rfft -> gate multiply -> (+memory) -> irfft -> slice (spectre.py:506, :542-553): `fft_amd.functional.
  spectral_mix`, one fused gfx950 kernel through the C ABI.  HIP device only; CPU tensors raise.

Training: the spectral mix is an autograd.Function; dV runs the same kernels with the conjugated filter and
dgate a spectrum-product reduction kernel.
"""
from __future__ import annotations

import math
import warnings
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .functional import (spectral_gate_backward, spectral_gate_fused, spectral_memory_grad, spectral_mix, spectral_mix_backward,
                         wavelet_gate_grad, wavelet_refine)

try:  # optional, exactly as the reference treats it (spectre.py:10-14)
    import torch_dct as _dct
except ImportError:  # pragma: no cover - not installed in the build image
    _dct = None


# --------------------------------------------------------------------------------------------------
# gate-producer pieces (host logic, plain torch)
# --------------------------------------------------------------------------------------------------
def resample_complex(anchors: torch.Tensor, size: int, mode: str = "cubic") -> torch.Tensor:
    """(B, G, K) complex anchors -> (B, G, size) complex, endpoints aligned.

    "cubic" follows spectre.py:38-61: the real/imag planes form a 2-channel image of height 1 that
    `grid_sample(mode='bicubic', padding_mode='border', align_corners=True)` samples on an even grid.
    "linear"/"nearest" follow spectre.py:75-92 (`F.interpolate`).
    """
    B, G, K = anchors.shape
    if mode == "cubic":
        planes = torch.stack((anchors.real, anchors.imag), dim=1).reshape(B * G, 2, 1, K)
        xs = torch.linspace(-1, 1, size, device=anchors.device)
        grid = torch.stack((xs, torch.zeros_like(xs)), dim=-1).view(1, 1, size, 2).expand(B * G, 1, size, 2)
        up = F.grid_sample(planes, grid, mode="bicubic", padding_mode="border", align_corners=True)
        return torch.complex(up[:, 0, 0, :], up[:, 1, 0, :]).view(B, G, size)
    if mode not in ("linear", "nearest"):
        raise AssertionError(f"Unsupported interpolation mode: {mode}")
    kw = {"align_corners": True} if mode == "linear" else {}
    re = F.interpolate(anchors.real.reshape(B * G, 1, K), size=size, mode=mode, **kw)
    im = F.interpolate(anchors.imag.reshape(B * G, 1, K), size=size, mode=mode, **kw)
    return torch.complex(re.squeeze(1), im.squeeze(1)).view(B, G, size)


def interp_complex_1d(x: torch.Tensor, size: int, mode: str = "linear") -> torch.Tensor:
    """The reference's name and defaults for `resample_complex` (spectre.py:26-30; its layers call it with mode="cubic", :526-528)."""
    return resample_complex(x, size, mode)


def complex_conv1d(x: torch.Tensor, kernel: torch.Tensor, padding: int) -> torch.Tensor:
    """Complex circular cross-correlation along the last axis — same name, arguments and result as spectre.py:334-395:
    `out[..., l] = sum_t kernel[t] * x[..., (l + t - padding) mod L]`, kernel (2 * padding + 1,) complex.

    The reference runs four real `conv1d`s on circularly padded planes (its `padding_mode='circular'` branch cannot be taken: `F.conv1d` has
    no such argument, so the padded fallback :384-391 is what executes); here the four are ONE real conv1d with two input and two output
    channels (weight [[k_r, -k_i], [k_i, k_r]]) on the padded (re, im) planes — O(B * G * K) work on the anchors, differentiable by autograd."""
    *batch_shape, L = x.shape
    planes = torch.stack((x.real, x.imag), dim=-2).reshape(-1, 2, L)
    planes = F.pad(planes, (padding, padding), mode="circular")
    kr, ki = kernel.real, kernel.imag
    weight = torch.stack((torch.stack((kr, -ki)), torch.stack((ki, kr))))          # (out = 2, in = 2, K)
    y = F.conv1d(planes, weight.to(planes.dtype))
    return torch.complex(y[:, 0], y[:, 1]).reshape(*batch_shape, L)


class ComplexModReLU(nn.Module):
    """z -> z * relu(|z| + b) / sqrt(|z|^2 + eps^2), one real bias per element (spectre.py:95-121)."""

    def __init__(self, num_features: int):
        super().__init__()
        self.bias = nn.Parameter(torch.full((num_features,), -0.1))
        self.register_buffer("eps", torch.tensor(1e-4))
        self.eps_value = 1e-4          # host copy for the fused gate kernel (reading the buffer would synchronise)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self.eps_value = float(self.eps)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        mag = torch.abs(z)
        denom = torch.sqrt(mag.square() + self.eps.square())
        return z * (F.relu(mag + self.bias) / denom)


class DCTPooling(nn.Module):
    """Mean of the first `dct_components` DCT coefficients along the sequence (spectre.py:136-156).

    Needs the optional `torch_dct` package; without it the reference warns and mean-pools, and so do we."""

    def __init__(self, embed_dim: int, dct_components: int = 64):
        super().__init__()
        self.embed_dim = embed_dim
        self.dct_components = dct_components

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _dct is not None:
            return _dct.dct(x.transpose(1, 2))[:, :, : self.dct_components].mean(dim=2)
        warnings.warn("DCT pooling unavailable, falling back to mean pooling. "
                      "Consider installing torch_dct or re-tuning hyperparameters.")
        return x.mean(dim=1)


class AttentionPooling(nn.Module):
    """softmax(w2(gelu(w1 x))) weighted sum over the sequence (spectre.py:159-172)."""

    def __init__(self, embed_dim: int, hidden_dim: int = 256):
        super().__init__()
        self.w1 = nn.Linear(embed_dim, hidden_dim)
        self.w2 = nn.Linear(hidden_dim, 1)
        self.activation = nn.GELU()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w = F.softmax(self.w2(self.activation(self.w1(x))), dim=1)
        return (x * w).sum(dim=1)


class MeanPool(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.mean(dim=1)


# --------------------------------------------------------------------------------------------------
# the fused op as an autograd node (forward only)
# --------------------------------------------------------------------------------------------------
class _SpectralMixFn(torch.autograd.Function):
    """autograd node of the fused mix: dV reuses the forward kernels with conj(gate), dgate is a spectrum product
    reduced over the channels of each group (fft_amd.functional.spectral_mix_backward)."""

    @staticmethod
    def forward(ctx, V, gate, memory_fft, n_fft):
        ctx.save_for_backward(V, gate)
        ctx.n_fft = n_fft
        return spectral_mix(V, gate, None if memory_fft is None else memory_fft.detach(), n_fft)

    @staticmethod
    @torch.autograd.function.once_differentiable      # the gradients come from C-ABI launches: no graph behind them, so double backward must raise
    def backward(ctx, grad_out):
        V, gate = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        dv, dgate = spectral_mix_backward(V, gate, grad_out, ctx.n_fft,
                                          need_dv=ctx.needs_input_grad[0], need_dgate=ctx.needs_input_grad[1])
        # memory_fft is frozen in the reference (spectre.py:961) but a caller may un-freeze it: the spectrum is added in front of the
        # irfft (:548-551), so its gradient is the irfft's adjoint of the batch sum of grad_out — one (N, D) reduction + one small rfft launch
        dmem = spectral_memory_grad(grad_out, ctx.n_fft) if ctx.needs_input_grad[2] else None
        return dv, dgate, dmem, None


class _SpectralGateFn(torch.autograd.Function):
    """autograd node of the fused gate producer (cubic resample -> modReLU -> positional phase): one launch forward, two backward
    (fft_amd.functional.spectral_gate_fused / spectral_gate_backward) instead of the ~12 + ~20 ATen launches of spectre.py:518-536."""

    @staticmethod
    def forward(ctx, anchors, bias, pos_phase, eps, size):
        ctx.save_for_backward(anchors, bias, pos_phase)
        ctx.eps, ctx.size = eps, size
        return spectral_gate_fused(anchors, bias, eps, size, pos_phase)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_gate):
        anchors, bias, pos_phase = ctx.saved_tensors
        da, db, dp = spectral_gate_backward(anchors, bias, ctx.eps, ctx.size, pos_phase, grad_gate.to(torch.complex64),
                                            need_dphase=pos_phase is not None and ctx.needs_input_grad[2])
        return (da if ctx.needs_input_grad[0] else None, db if ctx.needs_input_grad[1] else None, dp, None, None)


# --------------------------------------------------------------------------------------------------
# the layer
# --------------------------------------------------------------------------------------------------
class SpectreHead(nn.Module):
    """Frequency-domain token mixer for one head — same surface as spectre.py:400-557."""

    fold_mean_pooling = True       # on HIP devices, pool x before W_q when the pooling is the plain mean (see spectral_gate)
    fused_gate_autograd = True     # on HIP devices, run the gate producer tail as one autograd node (False: the reference's PyTorch ops)

    def __init__(self, embed_dim: int, fft_size: int, *, num_groups: int = 4, num_buckets: Optional[int] = None,
                 d_gate: int = 256, use_toeplitz: bool = False, toeplitz_bw: int = 4, dropout_p: float = 0.0,
                 pooling_type: str = "dct"):
        super().__init__()
        assert embed_dim % num_groups == 0, "embed_dim must be divisible by num_groups"
        self.d = embed_dim
        self.n_fft = fft_size
        self.G = num_groups
        self.d_g = embed_dim // num_groups
        self.F_half = fft_size // 2 + 1
        self.B = max(4, num_buckets or int(math.sqrt(self.F_half)))     # anchors per group

        self.W_q = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_v = nn.Linear(embed_dim, embed_dim, bias=False)
        self.gate_mlp = nn.Sequential(nn.Linear(embed_dim, d_gate), nn.GELU(), nn.Linear(d_gate, 2 * self.B * self.G))
        self.q_norm = nn.LayerNorm(embed_dim)
        self.modrelu = ComplexModReLU(self.F_half * self.G)
        if pooling_type == "dct":
            self.pooling = DCTPooling(embed_dim)
        elif pooling_type == "attention":
            self.pooling = AttentionPooling(embed_dim)
        else:
            self.pooling = MeanPool()

        # Toeplitz option (spectre.py:451-474, :519-521): a learnt complex kernel of 2 * bw + 1 taps, circularly correlated with the anchors along the
        # bucket axis and added to them.  The reference's constructor cannot build it on current PyTorch (`self.toeplitz_kernel = None` followed by
        # `register_parameter('toeplitz_kernel', None)` raises KeyError, :453-457); this one builds what `_reset_parameters` (:464-474) intends —
        # same parameter name, shape, dtype and initial scale — and the forward is the reference's own statement (pinned by running the reference's
        # forward with the parameter attached by hand: tests/golden/make_golden.py `case_toeplitz`).
        self.use_toeplitz = use_toeplitz
        if use_toeplitz:
            self.toeplitz_kernel = nn.Parameter(torch.randn(2 * toeplitz_bw + 1, dtype=torch.cfloat) / math.sqrt(2 * toeplitz_bw + 1))
        else:
            self.toeplitz_kernel = None
        self.toeplitz_bw = toeplitz_bw
        self.dropout = nn.Dropout(dropout_p) if dropout_p > 0 else nn.Identity()

    # ---- host logic: everything up to the filter the kernel consumes (spectre.py:502-503, :511-536) --
    def spectral_gate(self, x: torch.Tensor, pos_phase: Optional[torch.Tensor] = None, *, v_out: Optional[torch.Tensor] = None,
                      with_value: bool = True) -> Tuple[Optional[torch.Tensor], torch.Tensor, torch.Tensor]:
        """Returns (V (B,N,d), gate (B,G,F_half) complex64, q_pool (B,d)).  `v_out` (inference only): a (B,N,d) view the value
        projection is written into — e.g. this head's channel slice of a multi-head buffer — instead of a fresh tensor.
        `with_value=False`: the caller projects the values itself (the multi-head layer does it for all heads at once); V is None."""
        Bsz, N, d = x.shape
        assert d == self.d
        if not with_value:
            V = None
        elif v_out is not None:
            if torch.is_grad_enabled() and (x.requires_grad or self.W_v.weight.requires_grad):
                raise RuntimeError("spectral_gate(v_out=...) writes the value projection in place and records no graph: inference only")
            V = torch.matmul(x, self.W_v.weight.t(), out=v_out)          # W_v has no bias (spectre.py:428)
        else:
            V = self.W_v(x)
        if self.fold_mean_pooling and x.is_cuda and isinstance(self.pooling, MeanPool):
            # mean_n(W_q x_n) = W_q(mean_n x_n): W_q has no bias and Q feeds nothing but the pooling (spectre.py:502,
            # :511-512), so the (B,N,d) x (d,d) GEMM and the pass over Q collapse into a row mean and a (B,d) GEMM —
            # half of the layer's GEMM time.  Same value up to fp32 summation order.
            q_pool = self.q_norm(self.W_q(x.mean(dim=1)))
        else:
            q_pool = self.q_norm(self.pooling(self.W_q(x)))
        # .float(): under autocast the MLP returns bf16 / fp16, which view_as_complex refuses (the reference raises there: it is fp32-only);
        # the filter is complex64 either way, the mix then runs on the autocast dtype's rows (bf16 storage, fp32 arithmetic)
        anchors = torch.view_as_complex(self.gate_mlp(q_pool).float().view(Bsz, self.G, self.B, 2))
        if self.use_toeplitz:                                                   # spectre.py:519-521
            anchors = anchors + complex_conv1d(anchors, self.toeplitz_kernel, self.toeplitz_bw)
        wants_graph = torch.is_grad_enabled() and (anchors.requires_grad or self.modrelu.bias.requires_grad or
                                                   (pos_phase is not None and pos_phase.requires_grad))
        if x.is_cuda and anchors.dtype == torch.complex64 and not wants_graph:
            # inference: resample -> modReLU -> phase as one HIP launch (row N2); training keeps the ops autograd sees (also when
            # only modrelu.bias or a learnable pos_phase needs a gradient: the fused launch builds no graph)
            return V, spectral_gate_fused(anchors, self.modrelu.bias.detach(), self.modrelu.eps_value, self.F_half, pos_phase), q_pool
        if (self.fused_gate_autograd and x.is_cuda and anchors.dtype == torch.complex64 and self.modrelu.bias.dtype == torch.float32
                and (pos_phase is None or pos_phase.is_complex())):
            # training: the same launch, as an autograd node with its own two-launch backward (row N2 under autograd)
            return V, _SpectralGateFn.apply(anchors, self.modrelu.bias, pos_phase, self.modrelu.eps_value, self.F_half), q_pool
        gate = resample_complex(anchors, self.F_half, mode="cubic")
        gate = self.modrelu(gate.reshape(Bsz, -1)).view_as(gate)
        if pos_phase is not None:
            gate = gate * pos_phase.unsqueeze(1 if pos_phase.dim() == 2 else 0)
        return V, gate, q_pool

    def forward(self, x: torch.Tensor, pos_phase: Optional[torch.Tensor] = None, return_q_pool: bool = False,
                memory_fft: Optional[torch.Tensor] = None):
        V, gate, q_pool = self.spectral_gate(x, pos_phase)
        gate = gate.to(torch.complex64)
        if memory_fft is not None:
            memory_fft = memory_fft.to(torch.complex64)
        if torch.is_grad_enabled() and (V.requires_grad or gate.requires_grad or (memory_fft is not None and memory_fft.requires_grad)):
            mixed = _SpectralMixFn.apply(V, gate, memory_fft, self.n_fft)
        else:
            mixed = spectral_mix(V, gate, memory_fft, self.n_fft)          # spectre.py:506, :542-553
        result = self.dropout(mixed)
        if return_q_pool:
            return result, q_pool
        return result

    @torch.no_grad()
    def decode_step(self, q_t: torch.Tensor, v_t: torch.Tensor, cache) -> torch.Tensor:
        """Incremental generation update for one head, batch size 1 (spectre.py:564-611): q_t, v_t (d,) and a
        `fft_amd.PrefixFFTCache`; returns the mixed vector (d,) of the current time step."""
        from .decode import head_decode_step
        return head_decode_step(self, q_t, v_t, cache)


# --------------------------------------------------------------------------------------------------
# multi-head wrapper
# --------------------------------------------------------------------------------------------------
class _MultiHeadValueFn(torch.autograd.Function):
    """Value projections of all heads (spectre.py:503 inside the per-head loop of :712-719) written straight into the channel
    slices of ONE (B, N, D) tensor: V[..., h] = x[..., h] @ W_v^h.T — H GEMMs with strided operands, no concatenation pass
    over the activations in either direction (dx is assembled the same way)."""

    @staticmethod
    def forward(ctx, x, *weights):
        hd = weights[0].shape[0]
        V = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        for i, w in enumerate(weights):
            torch.matmul(x[..., i * hd:(i + 1) * hd], w.t(), out=V[..., i * hd:(i + 1) * hd])      # W_v has no bias (spectre.py:428)
        ctx.save_for_backward(x, *weights)
        return V

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dV):
        x, *weights = ctx.saved_tensors
        hd = weights[0].shape[0]
        dx = torch.empty(x.shape, dtype=x.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        dws = []
        for i, w in enumerate(weights):
            sl = slice(i * hd, (i + 1) * hd)
            dVi = dV[..., sl]
            if dx is not None:
                torch.matmul(dVi, w, out=dx[..., sl])
            dws.append(dVi.reshape(-1, hd).t() @ x[..., sl].reshape(-1, hd) if ctx.needs_input_grad[1 + i] else None)
        return (dx, *dws)


class _WaveletRefineFn(torch.autograd.Function):
    """autograd node of the refinement launch: `v + (R(v).detach() * gate) * on_mask` (spectre.py:884-886) — d/dv is the identity, d/dgate
    one reduction launch over the round trip the forward launch left behind for the switched-on elements."""

    @staticmethod
    def forward(ctx, v, gate, on_mask, inplace):
        want_ref = ctx.needs_input_grad[1]
        out, vref = wavelet_refine(v, gate, on_mask, inplace=inplace, want_ref=want_ref)
        if inplace:
            ctx.mark_dirty(v)
        ctx.save_for_backward(on_mask, *([vref] if want_ref else []))
        ctx.gate_dtype = gate.dtype
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        on_mask, *ref = ctx.saved_tensors
        dgate = wavelet_gate_grad(grad_out, ref[0], on_mask).to(ctx.gate_dtype) if ref else None
        return grad_out, dgate, None, None


class WaveletRefinement(nn.Module):
    """The optional refinement between the heads' concatenation and `out_proj` — same surface as spectre.py:819-887 (`on_rate`, `gate_mlp`,
    `forward(v, q_pool)`), same coin flip per batch element and forward pass (`torch.rand(B, 1, 1, device=v.device) < on_rate`, :841).

    The reference loops over the batch in Python, transposes every switched-on element, runs its Haar analysis / synthesis pair (which, because
    of the analysis' one-sample circular pad, is a fixed linear operator R != identity along the sequence — SURVEY.md section 2 row 10) and
    stacks the results; here that is ONE launch (`fft_amd.functional.wavelet_refine`, kernel_wavelet.h) that reads the mask on the device, so
    there is no `on_mask.any()` round trip to the host (:845) and switched-off elements cost nothing.  Like the reference, only power-of-two
    sequence lengths work (its synthesis fails with a size mismatch otherwise, :271) — here that is a ValueError whenever on_rate > 0, not
    only when the coin happens to switch an element on.  Gradients: identity to v, through `gate_mlp` to q_pool (the round trip is detached, :884).
    """

    def __init__(self, embed_dim: int, on_rate: float = 0.1):
        super().__init__()
        self.on_rate = on_rate
        self.gate_mlp = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.SiLU(), nn.Linear(embed_dim, embed_dim), nn.Sigmoid())
        self.forced_mask: Optional[torch.Tensor] = None      # tests / reproducing a reference run: use this (B,) bool mask instead of the draw

    def _draw(self, B: int, device) -> torch.Tensor:
        drawn = torch.rand(B, 1, 1, device=device) < self.on_rate          # always drawn, like the reference: the generator advances equally
        return drawn.view(B) if self.forced_mask is None else self.forced_mask.to(device=device, dtype=torch.bool).view(B)

    def forward(self, v: torch.Tensor, q_pool: torch.Tensor, *, inplace: bool = False) -> torch.Tensor:
        B, N, d = v.shape
        on_mask = self._draw(B, v.device)
        if self.on_rate <= 0.0 and self.forced_mask is None:
            return v                                                       # never on: the reference's early exit (:845-846)
        if N & (N - 1):
            raise ValueError(f"WaveletRefinement: sequence length {N} is not a power of two — the reference's Haar synthesis raises a size "
                             "mismatch for it as soon as a batch element is switched on (spectre.py:271); use wavelet_on_rate=0.0")
        gate = self.gate_mlp(q_pool)
        if torch.is_grad_enabled() and (v.requires_grad or gate.requires_grad):
            inplace = inplace and not (v.is_leaf and v.requires_grad) and v._base is None
            return _WaveletRefineFn.apply(v, gate, on_mask, inplace)
        return wavelet_refine(v, gate, on_mask, inplace=inplace)[0]


class SpectreMultiHead(nn.Module):
    """Several SpectreHeads side by side, the wavelet refinement and the output projection — same surface as spectre.py:660-726.

    The reference loops over heads and concatenates their outputs (`:712-719`); here every head's fused mix writes
    straight into its channel slice of one (B, N, embed_dim) buffer (strided output views of the C ABI), so the
    concatenation pass over the activations disappears.  The refinement (`:724`, default `wavelet_on_rate=0.1` like the reference —
    stochastic per batch element, active in eval() too) is one more launch, in place on that buffer (`WaveletRefinement`).
    """

    def __init__(self, embed_dim: int, num_heads: int, n_fft: int, d_gate: int = 256, use_toeplitz: bool = False,
                 dropout_p: float = 0.0, pooling_type: str = "dct", num_groups: int = 4,
                 num_buckets: Optional[int] = None, wavelet_on_rate: float = 0.1):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.heads = nn.ModuleList([
            SpectreHead(self.head_dim, fft_size=n_fft, d_gate=d_gate, use_toeplitz=use_toeplitz, dropout_p=dropout_p,
                        pooling_type=pooling_type, num_groups=num_groups, num_buckets=num_buckets)
            for _ in range(num_heads)])
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=False)
        self.wavelet_refinement = WaveletRefinement(embed_dim, on_rate=wavelet_on_rate)

    fused_autograd = True          # under autograd: one value-projection node + ONE spectral-mix node for all heads (False: the reference's loop)

    def _forward_per_head(self, x, pos_phase, memory_fft):
        """The reference's own structure (spectre.py:712-724): per-head modules, concatenated, refined.  Kept for autocast and as the
        comparison path of the tests."""
        chunks = torch.chunk(x, self.num_heads, dim=-1)
        mems = torch.chunk(memory_fft, self.num_heads, dim=-1) if memory_fft is not None else [None] * self.num_heads
        pairs = [h(c, pos_phase, return_q_pool=True, memory_fft=m) for h, c, m in zip(self.heads, chunks, mems)]
        mixed = torch.cat([m for m, _ in pairs], dim=-1)
        mixed = self.wavelet_refinement(mixed, torch.cat([q for _, q in pairs], dim=-1), inplace=True)      # `mixed` is the cat's own buffer
        return self.out_proj(mixed)

    def forward(self, x: torch.Tensor, pos_phase: Optional[torch.Tensor] = None, memory_fft: Optional[torch.Tensor] = None):
        chunks = torch.chunk(x, self.num_heads, dim=-1)
        needs_graph = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())
                                                   or (pos_phase is not None and pos_phase.requires_grad)
                                                   or (memory_fft is not None and memory_fft.requires_grad))
        p_drop = self.heads[0].dropout.p if isinstance(self.heads[0].dropout, nn.Dropout) else 0.0
        if needs_graph:
            if not self.fused_autograd or torch.is_autocast_enabled():
                return self._forward_per_head(x, pos_phase, memory_fft)
            # Training (row N3 under autograd): the per-head loop + torch.cat of the reference moves the whole activation once more in
            # each direction — as many bytes as the mix itself.  Here: one node projects every head's values into its channel slice of
            # one (B, N, D) tensor, the heads' gates are stacked to (B, H*G, F), and ONE spectral-mix node covers all heads (its
            # backward = one dV launch + one dgate launch over H*G groups).  Dropout (same p in every head, spectre.py:689) acts on
            # the fused tensor: the same distribution as per-head masks, drawn in one call.
            V = _MultiHeadValueFn.apply(x, *[h.W_v.weight for h in self.heads])
            gates, pools = [], []
            for h, c in zip(self.heads, chunks):
                _, gate, q_pool = h.spectral_gate(c, pos_phase, with_value=False)
                gates.append(gate.to(torch.complex64))
                pools.append(q_pool)
            mem = None if memory_fft is None else memory_fft.to(torch.complex64)
            mixed = _SpectralMixFn.apply(V, torch.cat(gates, dim=1), mem, self.heads[0].n_fft)
            if p_drop > 0.0 and self.training:
                mixed = F.dropout(mixed, p_drop, True)
            mixed = self.wavelet_refinement(mixed, torch.cat(pools, dim=-1), inplace=True)
            return self.out_proj(mixed)
        if p_drop > 0.0 and self.training:
            return self._forward_per_head(x, pos_phase, memory_fft)
        # One launch for all heads (row N3): every head's value projection lands in its channel slice of one (B, N, D) tensor,
        # the heads' gates are stacked to (B, H*G, F) — channel c of the full tensor belongs to gate row c // d_g exactly as in
        # the per-head calls — and memory_fft is used un-chunked.  Replaces the reference's per-head loop + torch.cat (:712-719).
        hd = self.head_dim
        V = torch.empty(x.shape[0], x.shape[1], x.shape[2], dtype=x.dtype, device=x.device)
        gates, pools = [], []
        for i, (h, c) in enumerate(zip(self.heads, chunks)):
            _, gate, q_pool = h.spectral_gate(c, pos_phase, v_out=V[:, :, i * hd:(i + 1) * hd])
            gates.append(gate.to(torch.complex64))
            pools.append(q_pool)
        mixed = spectral_mix(V, torch.cat(gates, dim=1), None if memory_fft is None else memory_fft.to(torch.complex64), self.heads[0].n_fft)
        mixed = self.wavelet_refinement(mixed, torch.cat(pools, dim=-1), inplace=True)
        return self.out_proj(mixed)


# --------------------------------------------------------------------------------------------------
# transformer block (the outermost caller of the path, SURVEY.md section 8(b) "Callers")
# --------------------------------------------------------------------------------------------------
class SpectreBlock(nn.Module):
    """Pre-norm residual block around the multi-head mix — same surface as spectre.py:892-982 (constructor keywords, `ln1` / `mix` /
    `ln2` / `mlp` / `memory_fft` names, `forward(x)`), so a reference block's `state_dict()` loads unchanged.

    What reaches the hot path from here is the block's frozen spectral memory (spectre.py:946-965): a `(mem_bins, embed_dim)` complex
    parameter that the reference zero-pads to `n_fft // 2 + 1` bins on EVERY forward (`:973-977`) and chunks per head (`:706-707`).
    Here the padded spectrum is built once per parameter version and handed un-chunked to the one fused launch over all heads (row a4 +
    row N3).  LayerNorm, the MLP and the residual adds are stock PyTorch-ROCm ops (GEMM-bound).
    `wavelet_on_rate`: the stochastic refinement of the layer inside (`WaveletRefinement`; the reference's default 0.1).
    """

    def __init__(self, embed_dim: int, num_heads: int, n_fft: int, mlp_ratio: int = 4, d_gate: int = 256, use_toeplitz: bool = False,
                 dropout_p: float = 0.0, pooling_type: str = "dct", num_groups: int = 4, num_buckets: Optional[int] = None,
                 wavelet_on_rate: float = 0.1, memory_size: int = 0):
        super().__init__()
        self.ln1 = nn.LayerNorm(embed_dim)
        self.mix = SpectreMultiHead(embed_dim, num_heads, n_fft, d_gate=d_gate, use_toeplitz=use_toeplitz, dropout_p=dropout_p,
                                    pooling_type=pooling_type, num_groups=num_groups, num_buckets=num_buckets,
                                    wavelet_on_rate=wavelet_on_rate)
        self.ln2 = nn.LayerNorm(embed_dim)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, mlp_ratio * embed_dim), nn.GELU(), nn.Linear(mlp_ratio * embed_dim, embed_dim))
        self.full_freq_bins = n_fft // 2 + 1
        if memory_size > 0:
            # memory_size == 1 means "all bins", > 1 that many (truncated) bins — spectre.py:949
            bins = min(memory_size, self.full_freq_bins) if memory_size > 1 else self.full_freq_bins
            self.memory_fft = nn.Parameter(torch.randn(bins, embed_dim, dtype=torch.cfloat) / math.sqrt(embed_dim), requires_grad=False)
            self.memory_freq_bins = bins
        else:
            self.memory_fft = None
        self._mem_padded = None            # (key, tensor): the (F, embed_dim) spectrum the kernel reads

    def _memory_spectrum(self) -> Optional[torch.Tensor]:
        m = self.memory_fft
        if m is None:
            return None
        if m.requires_grad and torch.is_grad_enabled():            # un-frozen by the caller: keep it in the graph, pad per call like the reference
            return F.pad(m, (0, 0, 0, self.full_freq_bins - m.shape[0])) if m.shape[0] < self.full_freq_bins else m
        key = (m.data_ptr(), m._version, m.device, m.shape[0])
        if self._mem_padded is None or self._mem_padded[0] != key:
            full = m.detach().to(torch.complex64)
            if full.shape[0] < self.full_freq_bins:
                full = F.pad(full, (0, 0, 0, self.full_freq_bins - full.shape[0]))
            self._mem_padded = (key, full.contiguous())
        return self._mem_padded[1]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x + self.mix(self.ln1(x), memory_fft=self._memory_spectrum())
        return x + self.mlp(self.ln2(x))
