"""spectral_mix — the fused rfft -> gate -> (+memory) -> irfft forward on MI355X.

Python face of the C ABI in include/spectre_hip.h; replaces these statements of the reference layer
(`/root/reference/spectre.py`):

    V_fft  = torch.fft.rfft(V, n=n_fft, dim=1)                                   # :506
    mixed  = gate.permute(0,2,1).repeat_interleave(d_g, -1) * V_fft              # :542-545
    mixed  = mixed + memory_fft.unsqueeze(0)                                     # :548-549 (optional)
    out    = torch.fft.irfft(mixed, n=n_fft, dim=1)[:, :N]                       # :551-553

PyTorch is used for device memory and streams only.  There is no eager/CPU fallback: CPU tensors,
unsupported dtypes or a missing shared library raise.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _native

_DT = {torch.float32: _native.F32, torch.bfloat16: _native.BF16}


def _args(V, gate, mem, n_fft, out, algo):
    if V.dim() != 3:
        raise ValueError(f"V must be (B, N, D), got {tuple(V.shape)}")
    if not V.is_cuda:
        raise RuntimeError("spectral_mix runs on a HIP device only (no CPU path): move V to cuda")
    if V.dtype not in _DT:
        raise TypeError(f"V dtype {V.dtype} unsupported (float32 or bfloat16)")
    B, N, D = V.shape
    F = n_fft // 2 + 1
    if gate.dim() != 3 or gate.shape[0] != B or gate.shape[2] != F:
        raise ValueError(f"gate must be (B={B}, G, F={F}) complex64, got {tuple(gate.shape)}")
    if gate.dtype != torch.complex64:
        raise TypeError(f"gate dtype {gate.dtype} unsupported (complex64)")
    G = gate.shape[1]
    if D % G:
        raise ValueError(f"D={D} not divisible by the number of gate channels G={G}")
    if gate.device != V.device or (mem is not None and mem.device != V.device):
        raise RuntimeError("V, gate and memory_fft must be on the same device")
    if V.stride(2) != 1:
        V = V.contiguous()
    gate = gate.contiguous()
    if mem is not None:
        if mem.dtype != torch.complex64 or tuple(mem.shape) != (F, D):
            raise ValueError(f"memory_fft must be (F={F}, D={D}) complex64, got {tuple(mem.shape)} {mem.dtype}")
        mem = mem.contiguous()
    a = _native.SpectreMixArgs()
    a.v = V.data_ptr()
    a.gate = gate.data_ptr()
    a.mem = mem.data_ptr() if mem is not None else None
    a.out = out.data_ptr()
    a.B, a.N_in, a.n_fft, a.D, a.G_tot = B, N, n_fft, D, G
    a.v_sb, a.v_sn = V.stride(0), V.stride(1)
    a.out_sb, a.out_sn = out.stride(0), out.stride(1)
    a.in_dtype, a.out_dtype = _DT[V.dtype], _DT[out.dtype]
    a.algo = _native.ALGO[algo]
    a.device = V.device.index if V.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(V.device).cuda_stream
    return a, (V, gate, mem, out)   # keep the (possibly re-laid-out) tensors alive until the launch is enqueued


def _empty_out(V, n_fft, out_dtype):
    B, N, D = V.shape
    return torch.empty((B, min(N, n_fft), D), dtype=out_dtype or V.dtype, device=V.device)


def spectral_mix(V: torch.Tensor, gate: torch.Tensor, memory_fft: Optional[torch.Tensor] = None,
                 n_fft: Optional[int] = None, *, out_dtype: Optional[torch.dtype] = None,
                 out: Optional[torch.Tensor] = None, algo: str = "auto") -> torch.Tensor:
    """out[b, n, c] = irfft(gate[b, c // d_g, :] * rfft(V[b, :, c], n_fft) + memory_fft[:, c], n_fft)[n].

    V (B, N, D) float32|bfloat16 (last dim unit stride; channel-chunk views with a larger row stride are
    fine), gate (B, G, n_fft//2+1) complex64, memory_fft (n_fft//2+1, D) complex64 or None.
    Returns (B, min(N, n_fft), D) in `out_dtype` (default: V.dtype).  Arithmetic is fp32.
    Asynchronous on the current stream.
    """
    lib = _native.load()
    if n_fft is None:
        n_fft = V.shape[1]
    if out is None:
        out = _empty_out(V, n_fft, out_dtype)
    elif out.stride(2) != 1 or tuple(out.shape) != (V.shape[0], min(V.shape[1], n_fft), V.shape[2]):
        raise ValueError("out has the wrong shape or a non-unit channel stride")
    a, keep = _args(V, gate, memory_fft, n_fft, out, algo)
    _native.check(lib.spectre_mix_fwd(ctypes.byref(a)), "spectre_mix_fwd")
    del keep
    return out


def spectral_mix_backward(V: torch.Tensor, gate: torch.Tensor, grad_out: torch.Tensor, n_fft: Optional[int] = None, *,
                          need_dv: bool = True, need_dgate: bool = True):
    """Gradients of `spectral_mix` w.r.t. V and gate for an upstream gradient `grad_out` (B, min(N, n_fft), D).

    dV = mix(grad_out, conj(gate)) (the forward kernels with the conjugated filter), zero for rows the forward
    truncated; dgate[b,g,k] = (w_k/n_fft) * sum_{c in g} conj(rfft(V)[k,c]) * rfft(grad_out)[k,c].  What autograd derives
    through `torch.fft.rfft/irfft` in the reference (spectre.py:506, :542-553).  Returns (dV or None, dgate or None).
    """
    lib = _native.load()
    if n_fft is None:
        n_fft = V.shape[1]
    if not V.is_cuda:
        raise RuntimeError("spectral_mix_backward runs on a HIP device only (no CPU path)")
    if V.dtype not in _DT or grad_out.dtype != V.dtype:
        raise TypeError("V and grad_out must share a dtype (float32 or bfloat16)")
    B, N, D = V.shape
    F = n_fft // 2 + 1
    n_out = min(N, n_fft)
    if tuple(grad_out.shape) != (B, n_out, D):
        raise ValueError(f"grad_out must be {(B, n_out, D)}, got {tuple(grad_out.shape)}")
    if gate.dim() != 3 or gate.shape[0] != B or gate.shape[2] != F or gate.dtype != torch.complex64 or D % gate.shape[1]:
        raise ValueError(f"gate must be (B={B}, G, F={F}) complex64 with G | D")
    if V.stride(2) != 1:
        V = V.contiguous()
    if grad_out.stride(2) != 1:
        grad_out = grad_out.contiguous()
    gate = gate.contiguous()
    G = gate.shape[1]
    dv = torch.empty((B, N, D), dtype=V.dtype, device=V.device) if need_dv else None
    dgate = torch.empty((B, G, F), dtype=torch.complex64, device=V.device) if need_dgate else None
    ws = None
    if need_dgate:
        # ALL scratch of the backward comes from torch's caching allocator (the library allocates nothing): the partial sums of the
        # gate gradient and, for n_fft = 12288 / 16384 / long Bluestein lengths, the spectra of the two-pass form
        ws = torch.empty(lib.spectre_mix_bwd_workspace_bytes(B, n_fft, D, G), dtype=torch.uint8, device=V.device)
    a = _native.SpectreMixBwdArgs()
    a.v, a.gate, a.dout = V.data_ptr(), gate.data_ptr(), grad_out.data_ptr()
    a.dv = dv.data_ptr() if dv is not None else None
    a.dgate = dgate.data_ptr() if dgate is not None else None
    a.workspace = ws.data_ptr() if ws is not None else None
    a.workspace_bytes = ws.numel() if ws is not None else 0
    a.B, a.N_in, a.n_fft, a.D, a.G_tot = B, N, n_fft, D, G
    a.v_sb, a.v_sn = V.stride(0), V.stride(1)
    a.dout_sb, a.dout_sn = grad_out.stride(0), grad_out.stride(1)
    if dv is not None:
        a.dv_sb, a.dv_sn = dv.stride(0), dv.stride(1)
    a.io_dtype = _DT[V.dtype]
    a.device = V.device.index if V.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(V.device).cuda_stream
    _native.check(lib.spectre_mix_bwd(ctypes.byref(a)), "spectre_mix_bwd")
    return dv, dgate


TILE_ORDERS = {"auto": 0, "static": 1, "tickets": 2, "pair": 3}       # SPECTRE_ORDER_* of include/spectre_hip.h


def spectral_memory_grad(grad_out: torch.Tensor, n_fft: int) -> torch.Tensor:
    """d/d(memory_fft) of `spectral_mix` for the upstream gradient `grad_out` (B, N_out, D): the (F, D) complex64 tensor torch's autograd
    returns for the `+ memory_fft` of spectre.py:548-549 followed by `irfft(..., n=n_fft)[:, :N]` (:551-553).

    The spectrum is shared by the batch and enters linearly, so the gradient is the adjoint of irfft applied to sum_b grad_out[b]:
    `w_k / n_fft * rfft(sum_b grad_out[b], n_fft)[k]` with w = 1 at DC (and at Nyquist for even n_fft), 2 elsewhere — the imaginary parts
    that irfft ignores at those bins get the zero gradient they have in the reference (rfft of a real sequence is real there).  The batch
    sum is a stock reduction; the transform is the library's own rfft launch (`spectre_rfft_fwd`, zero-padding rows N_out .. n_fft - 1)."""
    from .decode import rfft_prefill
    if grad_out.dim() != 3:
        raise ValueError(f"grad_out must be (B, N_out, D), got {tuple(grad_out.shape)}")
    spec = rfft_prefill(grad_out.sum(dim=0, dtype=torch.float32), n_fft)
    F = n_fft // 2 + 1
    w = torch.full((F, 1), 2.0 / n_fft, dtype=torch.float32, device=grad_out.device)
    w[0] = 1.0 / n_fft
    if n_fft % 2 == 0:
        w[-1] = 1.0 / n_fft
    return spec * w


def wavelet_refine(v: torch.Tensor, gate: torch.Tensor, on_mask: torch.Tensor, *, inplace: bool = False, want_ref: bool = False):
    """`v + (R(v) * gate[:, None, :]) * on_mask[:, None, None]` with R the Haar analysis / synthesis round trip of the reference's
    WaveletRefinement along the sequence (spectre.py:853-872, :884-886) — one launch, only the switched-on batch elements are read.

    v (B, N, D) f32|bf16 with N a power of two;  gate (B, D) (`gate_mlp(q_pool)`, :848);  on_mask (B) or (B, 1, 1) bool
    (`torch.rand(B, 1, 1) < on_rate`, :841).  inplace: write into v.  want_ref: also return R(v) for the switched-on elements (rows of
    the others are uninitialised) — the operand of `wavelet_gate_grad`.  Returns (out, vref | None)."""
    lib = _native.load()
    if not v.is_cuda:
        raise RuntimeError("wavelet_refine runs on a HIP device only (no CPU path)")
    if v.dim() != 3:
        raise ValueError(f"v must be (B, N, D), got {tuple(v.shape)}")
    if v.dtype not in _DT:
        raise TypeError(f"v dtype {v.dtype} unsupported (float32 or bfloat16)")
    B, N, D = v.shape
    if tuple(gate.shape) != (B, D):
        raise ValueError(f"gate must be (B={B}, D={D}), got {tuple(gate.shape)}")
    if on_mask.numel() != B or on_mask.dtype != torch.bool:
        raise ValueError(f"on_mask must hold B={B} booleans, got {tuple(on_mask.shape)} {on_mask.dtype}")
    if gate.device != v.device or on_mask.device != v.device:
        raise RuntimeError("v, gate and on_mask must be on the same device")
    if v.stride(2) != 1:
        if inplace:
            raise ValueError("wavelet_refine(inplace=True) needs unit stride along the channels")
        v = v.contiguous()
    gate = gate.detach().to(torch.float32).contiguous()
    mask = on_mask.reshape(B).contiguous()
    out = v if inplace else torch.empty((B, N, D), dtype=v.dtype, device=v.device)
    vref = torch.empty((B, N, D), dtype=v.dtype, device=v.device) if want_ref else None
    a = _native.SpectreWaveletArgs()
    a.v, a.out, a.vref = v.data_ptr(), out.data_ptr(), (vref.data_ptr() if vref is not None else None)
    a.mask, a.gate = mask.data_ptr(), gate.data_ptr()
    a.B, a.N, a.D = B, N, D
    a.v_sb, a.v_sn, a.out_sb, a.out_sn = v.stride(0), v.stride(1), out.stride(0), out.stride(1)
    a.ref_sb, a.ref_sn = (vref.stride(0), vref.stride(1)) if vref is not None else (0, 0)
    a.dtype = _DT[v.dtype]
    a.device = v.device.index if v.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(v.device).cuda_stream
    _native.check(lib.spectre_wavelet_refine(ctypes.byref(a)), "spectre_wavelet_refine")
    return out, vref


def wavelet_gate_grad(grad_out: torch.Tensor, vref: torch.Tensor, on_mask: torch.Tensor) -> torch.Tensor:
    """d/d(gate) of `wavelet_refine`: (B, D) f32 = on_mask * sum_n grad_out * R(v) (the round trip is detached in the reference,
    spectre.py:884, so d/dv is the identity and this is the only other gradient)."""
    lib = _native.load()
    if not grad_out.is_cuda:
        raise RuntimeError("wavelet_gate_grad runs on a HIP device only (no CPU path)")
    if grad_out.dim() != 3 or grad_out.shape != vref.shape or grad_out.dtype != vref.dtype or grad_out.dtype not in _DT:
        raise ValueError(f"grad_out and vref must be equal-shaped (B, N, D) f32|bf16 tensors, got {tuple(grad_out.shape)} {grad_out.dtype} "
                         f"and {tuple(vref.shape)} {vref.dtype}")
    if grad_out.stride(2) != 1:
        grad_out = grad_out.contiguous()
    B, N, D = grad_out.shape
    mask = on_mask.reshape(B).contiguous()
    dgate = torch.empty((B, D), dtype=torch.float32, device=grad_out.device)
    a = _native.SpectreWaveletGradArgs()
    a.dout, a.vref, a.mask, a.dgate = grad_out.data_ptr(), vref.data_ptr(), mask.data_ptr(), dgate.data_ptr()
    a.B, a.N, a.D = B, N, D
    a.d_sb, a.d_sn, a.ref_sb, a.ref_sn = grad_out.stride(0), grad_out.stride(1), vref.stride(0), vref.stride(1)
    a.dtype = _DT[grad_out.dtype]
    a.device = grad_out.device.index if grad_out.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(grad_out.device).cuda_stream
    _native.check(lib.spectre_wavelet_gate_grad(ctypes.byref(a)), "spectre_wavelet_gate_grad")
    return dgate


def set_tile_order(n_fft: int, order: str = "auto", device=None) -> None:
    """Tile order of the persistent kernels (n_fft = 4096, 3000, 3600, 3840) on `device` (C ABI `spectre_plan_set_tile_order`):
    "auto" (default: tickets, measured once per shape class, the static map only where it is at least 1 % faster), "static", "tickets"
    (pinned: no event calls on the launch path) or "pair" (measured per (V, out) pointer pair as well).  Forgets every decision so far."""
    lib = _native.load()
    if order not in TILE_ORDERS:
        raise ValueError(f"order must be one of {sorted(TILE_ORDERS)}")
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    _native.check(lib.spectre_plan_set_tile_order(int(dev), int(n_fft), TILE_ORDERS[order]), "spectre_plan_set_tile_order")


def get_tile_order(n_fft: int, device=None) -> str:
    lib = _native.load()
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    o = ctypes.c_int(-1)
    _native.check(lib.spectre_plan_get_tile_order(int(dev), int(n_fft), ctypes.byref(o)), "spectre_plan_get_tile_order")
    return {v: k for k, v in TILE_ORDERS.items()}[o.value]


def describe(V, gate, memory_fft=None, n_fft=None, *, out_dtype=None, algo="auto", out=None) -> str:
    """Name of the kernel `spectral_mix` would launch for these arguments.  `out`: the output tensor of the launch in question (the tile
    order of the persistent kernels is measured per shape class — under set_tile_order(..., "pair") per (V, out) pair: `order=auto` until
    it has been, then `auto:tickets (...)` / `auto:static (...)`)."""
    lib = _native.load()
    if n_fft is None:
        n_fft = V.shape[1]
    if out is None:
        out = _empty_out(V, n_fft, out_dtype)
    a, keep = _args(V, gate, memory_fft, n_fft, out, algo)
    buf = ctypes.create_string_buffer(512)
    _native.check(lib.spectre_mix_describe(ctypes.byref(a), buf, 512), "spectre_mix_describe")
    del keep
    return buf.value.decode()


def time_kernel(V, gate, memory_fft=None, n_fft=None, *, out_dtype=None, out=None, algo="auto",
                warmup: int = 3, iters: int = 10) -> float:
    """Average milliseconds per launch, measured with HIP events on the launch stream (bench.py)."""
    lib = _native.load()
    if n_fft is None:
        n_fft = V.shape[1]
    if out is None:
        out = _empty_out(V, n_fft, out_dtype)
    a, keep = _args(V, gate, memory_fft, n_fft, out, algo)
    ms = ctypes.c_float(0.0)
    _native.check(lib.spectre_mix_time(ctypes.byref(a), warmup, iters, ctypes.byref(ms)), "spectre_mix_time")
    del keep
    return float(ms.value)


def copy_probe(src: torch.Tensor, dst: torch.Tensor, seg_bytes: int = 0, *, tile_rows: int = 4096, mode: str = "copy",
               wgs_per_cu: int = 0, warmup: int = 2, iters: int = 5) -> float:
    """Milliseconds per launch of a PURE COPY of `src` into `dst` with the product kernels' access pattern (C ABI `spectre_probe_copy`;
    measurement only, used by bench.py for the driver-visible memory ceilings).  src / dst: contiguous (B, N, D) tensors of one dtype.
    seg_bytes 0 = dense copy; S = every workgroup moves S bytes of `tile_rows` consecutive rows (64 = the fp32 tile of the 4096 kernel);
    -1 = the plain non-persistent float4 copy (one workgroup per 4 KiB), -2 = the same with non-temporal accesses, -3 = hipMemcpyAsync."""
    lib = _native.load()
    if not (src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.shape == dst.shape and src.dtype == dst.dtype
            and src.dim() == 3):
        raise ValueError("copy_probe wants two contiguous (B, N, D) device tensors of the same shape and dtype")
    B, N, D = src.shape
    a = _native.SpectreProbeArgs()
    a.src, a.dst = src.data_ptr(), dst.data_ptr()
    a.rows, a.row_bytes = B * N, D * src.element_size()
    a.seg_bytes, a.tile_rows, a.mode, a.wgs_per_cu = int(seg_bytes), int(tile_rows), {"copy": 0, "load": 1, "store": 2}[mode], int(wgs_per_cu)
    a.device = src.device.index if src.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(src.device).cuda_stream
    ms = ctypes.c_float(0.0)
    _native.check(lib.spectre_probe_copy(ctypes.byref(a), warmup, iters, ctypes.byref(ms)), "spectre_probe_copy")
    return float(ms.value)


def empty_on_fast_allocation(shape, dtype=torch.float32, device="cuda", candidates: int = 4):
    """`torch.empty(shape)` on the fastest of `candidates` fresh allocations, by a store-only pass of the copy probe over each.

    On MI355X device allocations come in two classes; one takes stores ~19 % faster, and the class belongs to the allocation (LABNOTES.md
    section 5, round 3, item 7; tools/placement_classes.py).  For long-lived (B, N, D) activation buffers it is worth half a millisecond
    per candidate, once.  The losers go back to torch's caching allocator, which may hand them out again: allocate what must be fast
    first.  Returns (tensor, store_ms_of_every_candidate); tensors the dense probe cannot take (not 3-D, not a multiple of 256 KiB) come
    back as the first candidate with an empty list."""
    first = torch.empty(shape, dtype=dtype, device=device)
    if candidates < 2 or first.dim() != 3 or (first.numel() * first.element_size()) % (256 * 1024) or not first.is_cuda:
        return first, []
    cands = [first] + [torch.empty(shape, dtype=dtype, device=device) for _ in range(candidates - 1)]
    ms = [min(copy_probe(c, c, 0, mode="store", wgs_per_cu=w, warmup=3, iters=8) for w in (2, 4)) for c in cands]
    best = min(range(len(cands)), key=lambda i: ms[i])
    return cands[best], ms


def spectral_gate_fused(anchors: torch.Tensor, bias: torch.Tensor, eps: float, size: int,
                        pos_phase: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Cubic resample of (B, G, K) complex anchors to `size` bins -> complex modReLU -> optional positional phase,
    one HIP launch (C ABI `spectre_gate_fwd`; replaces /root/reference/spectre.py:518-524, :530-531, :534-536).

    bias: ComplexModReLU.bias, (G * size,) float32.  pos_phase: (size,), (1, size) or (B, size) complex.
    Forward only: no autograd graph is recorded (training keeps the PyTorch ops)."""
    lib = _native.load()
    if not anchors.is_cuda:
        raise RuntimeError("spectral_gate_fused runs on a HIP device only (no CPU path)")
    if anchors.dim() != 3 or anchors.dtype != torch.complex64:
        raise ValueError(f"anchors must be (B, G, K) complex64, got {tuple(anchors.shape)} {anchors.dtype}")
    B, G, K = anchors.shape
    if bias.dtype != torch.float32 or bias.numel() != G * size or bias.device != anchors.device:
        raise ValueError(f"bias must be ({G * size},) float32 on {anchors.device}")
    anchors = anchors.contiguous()
    bias = bias.contiguous()
    phase, phase_sb = None, 0
    if pos_phase is not None:
        phase = pos_phase.to(torch.complex64)
        if phase.device != anchors.device:
            raise RuntimeError("pos_phase must be on the same device as the anchors")
        if phase.dim() == 1 and phase.shape[0] == size:
            pass
        elif phase.dim() == 2 and phase.shape[1] == size and phase.shape[0] in (1, B):
            phase_sb = size if (phase.shape[0] == B and B > 1) else 0
        else:
            raise ValueError(f"pos_phase must be ({size},), (1, {size}) or ({B}, {size}), got {tuple(pos_phase.shape)}")
        phase = phase.contiguous()
    gate = torch.empty(B, G, size, dtype=torch.complex64, device=anchors.device)
    a = _native.SpectreGateArgs()
    a.anchors, a.bias, a.gate = anchors.data_ptr(), bias.data_ptr(), gate.data_ptr()
    a.phase = phase.data_ptr() if phase is not None else None
    a.B, a.G, a.K, a.F, a.phase_sb, a.eps = B, G, K, size, phase_sb, float(eps)
    a.device = anchors.device.index if anchors.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(anchors.device).cuda_stream
    _native.check(lib.spectre_gate_fwd(ctypes.byref(a)), "spectre_gate_fwd")
    return gate


def _gate_operands(anchors, bias, size, pos_phase):
    if not anchors.is_cuda:
        raise RuntimeError("the fused gate producer runs on a HIP device only (no CPU path)")
    if anchors.dim() != 3 or anchors.dtype != torch.complex64:
        raise ValueError(f"anchors must be (B, G, K) complex64, got {tuple(anchors.shape)} {anchors.dtype}")
    B, G, K = anchors.shape
    if bias.dtype != torch.float32 or bias.numel() != G * size or bias.device != anchors.device:
        raise ValueError(f"bias must be ({G * size},) float32 on {anchors.device}")
    phase, phase_sb = None, 0
    if pos_phase is not None:
        phase = pos_phase.to(torch.complex64)
        if phase.device != anchors.device:
            raise RuntimeError("pos_phase must be on the same device as the anchors")
        if phase.dim() == 1 and phase.shape[0] == size:
            pass
        elif phase.dim() == 2 and phase.shape[1] == size and phase.shape[0] in (1, B):
            phase_sb = size if (phase.shape[0] == B and B > 1) else 0
        else:
            raise ValueError(f"pos_phase must be ({size},), (1, {size}) or ({B}, {size}), got {tuple(pos_phase.shape)}")
        phase = phase.contiguous()
    return anchors.contiguous(), bias.contiguous(), phase, phase_sb


def spectral_gate_backward(anchors: torch.Tensor, bias: torch.Tensor, eps: float, size: int, pos_phase: Optional[torch.Tensor],
                           grad_gate: torch.Tensor, need_dphase: bool = False):
    """Gradients of `spectral_gate_fused` for an upstream gradient `grad_gate` (B, G, size) complex64: (d_anchors (B, G, K) complex64,
    d_bias (G * size,) float32, d_pos_phase in pos_phase's shape or None) — C ABI `spectre_gate_bwd`, two launches; what autograd
    derives through /root/reference/spectre.py:518-524, :530-531, :534-536."""
    lib = _native.load()
    anchors, bias, phase, phase_sb = _gate_operands(anchors, bias, size, pos_phase)
    B, G, K = anchors.shape
    if tuple(grad_gate.shape) != (B, G, size) or grad_gate.dtype != torch.complex64:
        raise ValueError(f"grad_gate must be ({B}, {G}, {size}) complex64")
    grad_gate = grad_gate.contiguous()
    d_anchors = torch.empty_like(anchors)
    d_bias = torch.empty(G * size, dtype=torch.float32, device=anchors.device)
    d_phase = torch.zeros_like(phase) if (need_dphase and phase is not None) else None
    ws = torch.empty(B * G * size, dtype=torch.complex64, device=anchors.device)
    a = _native.SpectreGateBwdArgs()
    a.anchors, a.bias, a.dgate, a.workspace = anchors.data_ptr(), bias.data_ptr(), grad_gate.data_ptr(), ws.data_ptr()
    a.phase = phase.data_ptr() if phase is not None else None
    a.danchors, a.dbias = d_anchors.data_ptr(), d_bias.data_ptr()
    a.dphase = d_phase.data_ptr() if d_phase is not None else None
    a.B, a.G, a.K, a.F, a.phase_sb, a.eps = B, G, K, size, phase_sb, float(eps)
    a.device = anchors.device.index if anchors.device.index is not None else torch.cuda.current_device()
    a.stream = torch.cuda.current_stream(anchors.device).cuda_stream
    _native.check(lib.spectre_gate_bwd(ctypes.byref(a)), "spectre_gate_bwd")
    if d_phase is not None and pos_phase is not None:
        d_phase = d_phase.reshape(pos_phase.shape).to(pos_phase.dtype)
    return d_anchors, d_bias, d_phase
