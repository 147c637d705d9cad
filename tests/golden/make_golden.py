#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run in the build container only (it needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference (`/root/reference/spectre.py`) ships no tests and no fixtures, so the
spectral-mix hot path (`spectre.py:506`, `:542-553`) is pinned here by importing the reference
module, driving `SpectreHead.forward` on seeded inputs and recording, per case,

    x            module input                         (B, N, d) f32
    sd/<name>    the module's state_dict              (so the drop-in module can load it)
    V            W_v(x), captured by a forward hook   (B, N, d) f32      -> kernel input
    gate         the filter that reaches `:542`       (B, G, F) c64      -> kernel input
                 (= modrelu output times pos_phase, `:531-536`)
    mem          memory_fft or absent                 (F, d) c64         -> kernel input
    out          SpectreHead.forward(x, ...)          (B, min(N,n_fft), d) f32

Every case is self-checked before it is written: `irfft(gate_bc * rfft(V) [+ mem])[:, :N]`
recomputed from the captured tensors must equal `out` bit for bit (SURVEY.md §4 identity 1).

Adversarial-gate cases (G6) and the bf16 case (G8) still run the reference's own forward: the
module's `modrelu` is swapped for a stub that returns a chosen gate and `W_v` is set to the
identity, so lines `:542-553` execute unmodified on inputs we control.

The files are data only: inputs and the outputs the reference produced for them.
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import spectre as ref  # noqa: E402  (the reference; never shipped)

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


class _FixedGate(torch.nn.Module):
    """Stub for `modrelu`: ignores its input and returns a preset (B, G*F) complex gate."""

    def __init__(self, gate_flat):
        super().__init__()
        self.g = gate_flat

    def forward(self, z):
        return self.g.reshape(z.shape)


def _capture(head, x, pos_phase=None, memory_fft=None):
    cap = {}
    h1 = head.W_v.register_forward_hook(lambda m, i, o: cap.__setitem__("V", o.detach().clone()))
    h2 = head.modrelu.register_forward_hook(lambda m, i, o: cap.__setitem__("g", o.detach().clone()))
    with torch.no_grad():
        out = head(x, pos_phase=pos_phase, memory_fft=memory_fft)
    h1.remove()
    h2.remove()
    B = x.shape[0]
    gate = cap["g"].reshape(B, head.G, head.F_half)
    if pos_phase is not None:  # spectre.py:534-536
        gate = gate * pos_phase.unsqueeze(1 if pos_phase.dim() == 2 else 0)
    return cap["V"], gate, out


def _selfcheck(V, gate, mem, n_fft, d_g, out):
    N = V.shape[1]
    vf = torch.fft.rfft(V, n=n_fft, dim=1)
    gb = gate.permute(0, 2, 1).repeat_interleave(d_g, dim=-1)
    mixed = gb * vf
    if mem is not None:
        mixed = mixed + mem.unsqueeze(0)
    y = torch.fft.irfft(mixed, n=n_fft, dim=1)[:, :N]
    assert y.shape == out.shape, (y.shape, out.shape)
    assert torch.equal(y, out), float((y - out).abs().max())


def _save(name, *, x, head, V, gate, out, mem=None, pos_phase=None, extra=None, with_sd=True):
    d = {"V": V.numpy(), "gate": gate.numpy().astype(np.complex64), "out": out.numpy(),
         "n_fft": np.int64(head.n_fft), "G": np.int64(head.G)}
    if with_sd:
        d["x"] = x.numpy()   # module-level cases only; fixed-gate cases have x == V (W_v = identity)
    if mem is not None:
        d["mem"] = mem.numpy().astype(np.complex64)
    if pos_phase is not None:
        d["pos_phase"] = pos_phase.numpy().astype(np.complex64)
    if with_sd:
        for k, v in head.state_dict().items():
            d["sd/" + k] = v.numpy()
    if extra:
        d.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"{name:28s} x{tuple(x.shape)} gate{tuple(gate.shape)} out{tuple(out.shape)}  {os.path.getsize(path) / 1e3:.0f} kB")


def _head(d, n_fft, G, seed, pooling="mean"):
    torch.manual_seed(seed)
    h = ref.SpectreHead(d, n_fft, num_groups=G, pooling_type=pooling).eval()
    return h


def case_module(name, B, N, d, n_fft, G, seed, *, pooling="mean", with_mem=False, phase_shape=None, with_sd=True):
    head = _head(d, n_fft, G, seed, pooling)
    g = torch.Generator().manual_seed(seed + 1000)
    x = torch.randn(B, N, d, generator=g)
    F = head.F_half
    mem = None
    if with_mem:
        mem = torch.complex(torch.randn(F, d, generator=g), torch.randn(F, d, generator=g)) * 0.5
    pp = None
    if phase_shape is not None:
        k = torch.arange(F, dtype=torch.float32)
        if phase_shape == "F":
            pp = torch.exp(1j * 2 * np.pi * k * 3.0 / n_fft)
        elif phase_shape == "1F":
            pp = torch.exp(1j * 2 * np.pi * k * 5.0 / n_fft).unsqueeze(0)
        elif phase_shape == "BF":
            shifts = torch.arange(B, dtype=torch.float32).unsqueeze(1) + 1.0
            pp = torch.exp(1j * 2 * np.pi * k.unsqueeze(0) * shifts / n_fft)
        pp = pp.to(torch.complex64)
    V, gate, out = _capture(head, x, pos_phase=pp, memory_fft=mem)
    _selfcheck(V, gate, mem, n_fft, head.d_g, out)
    _save(name, x=x, head=head, V=V, gate=gate, out=out, mem=mem, pos_phase=pp, with_sd=with_sd)


def case_fixed_gate(name, B, N, d, n_fft, G, seed, gate_fn, *, bf16=False, with_mem=False):
    """Run reference lines :542-553 on a gate we choose (W_v = identity so V == x)."""
    head = _head(d, n_fft, G, seed)
    with torch.no_grad():
        head.W_v.weight.copy_(torch.eye(d))
    g = torch.Generator().manual_seed(seed + 2000)
    x = torch.randn(B, N, d, generator=g)
    extra = {}
    if bf16:
        x = x.bfloat16().float()  # values exactly representable in bf16
    F = head.F_half
    gate = gate_fn(B, G, F, g).to(torch.complex64)
    head.modrelu = _FixedGate(gate.reshape(B, G * F))
    mem = None
    if with_mem:
        mem = torch.complex(torch.randn(F, d, generator=g), torch.randn(F, d, generator=g)) * 0.25
    V, gate_c, out = _capture(head, x, memory_fft=mem)
    assert torch.equal(V, x)  # identity projection is exact
    assert torch.equal(gate_c, gate)
    _selfcheck(V, gate, mem, n_fft, head.d_g, out)
    if bf16:
        extra["out_bf16_bits"] = out.bfloat16().view(torch.int16).numpy()
    _save(name, x=x, head=head, V=V, gate=gate, out=out, mem=mem, extra=extra, with_sd=False)


def case_backward(name, B, N, d, n_fft, G, seed):
    """Gradients the reference's autograd produces at the inputs of the hot path (V = W_v(x), gate = modReLU output)
    for a fixed upstream gradient dout: the N1 row of SURVEY.md section 8(f)."""
    head = _head(d, n_fft, G, seed)
    g = torch.Generator().manual_seed(seed + 3000)
    x = torch.randn(B, N, d, generator=g)
    cap = {}

    def keep(key):
        def hook(m, i, o):
            o.retain_grad()
            cap[key] = o
        return hook
    h1 = head.W_v.register_forward_hook(keep("V"))
    h2 = head.modrelu.register_forward_hook(keep("g"))
    out = head(x)
    h1.remove()
    h2.remove()
    dout = torch.randn(out.shape, generator=g)
    (out * dout).sum().backward()
    V, gflat = cap["V"], cap["g"]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, V=V.detach().numpy(), gate=gflat.detach().reshape(B, G, head.F_half).numpy().astype(np.complex64),
                        dout=dout.numpy(), dV=V.grad.numpy(), dgate=gflat.grad.reshape(B, G, head.F_half).numpy().astype(np.complex64),
                        out=out.detach().numpy(), n_fft=np.int64(n_fft), G=np.int64(G))
    print(f"{name:28s} backward x{tuple(x.shape)} dV{tuple(V.grad.shape)} dgate{(B, G, head.F_half)}  {os.path.getsize(path) / 1e3:.0f} kB")


def case_decode(name, d, n_fft, G, seed, L, T, *, with_mem=False):
    """Prefill + T single-token decode steps of one head (spectre.py:731-814 `PrefixFFTCache`, :564-611 `decode_step`,
    :614-655 `pruned_irfft_single`): row N4 of SURVEY.md section 8(f).  Inputs and everything the reference returned."""
    head = _head(d, n_fft, G, seed)
    g = torch.Generator().manual_seed(seed + 4000)
    Qp, Vp = torch.randn(L, d, generator=g), torch.randn(L, d, generator=g)
    q_seq, v_seq = torch.randn(T, d, generator=g), torch.randn(T, d, generator=g)
    cache = ref.PrefixFFTCache(n_fft, d, device=torch.device("cpu"))
    cache.prefill(Qp, Vp)
    extra = {"prefix_after_prefill": cache.prefix_fft.clone().numpy()}
    if with_mem:
        mem = torch.complex(torch.randn(n_fft // 2 + 1, d, generator=g), torch.randn(n_fft // 2 + 1, d, generator=g)) * 0.5
        cache.prefix_fft += mem                      # the usage the reference documents (spectre.py:736-741)
        extra["mem"] = mem.numpy()
    outs = torch.stack([head.decode_step(q_seq[i], v_seq[i], cache) for i in range(T)])
    d_out = {"Qp": Qp.numpy(), "Vp": Vp.numpy(), "q_seq": q_seq.numpy(), "v_seq": v_seq.numpy(), "outs": outs.numpy(),
             "prefix_final": cache.prefix_fft.numpy(), "sum_q_final": cache.sum_q.numpy(), "t_final": np.int64(cache.t),
             "n_fft": np.int64(n_fft), "G": np.int64(G)}
    d_out.update(extra)
    for k, v in head.state_dict().items():
        d_out["sd/" + k] = v.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d_out)
    print(f"{name:28s} decode L={L} T={T} d={d} n_fft={n_fft}  {os.path.getsize(path) / 1e3:.0f} kB")


def case_multihead(name, B, N, E, H, n_fft, G, seed, *, with_mem=False, with_phase=False, with_grad=False):
    """SpectreMultiHead.forward (spectre.py:660-726) with the stochastic wavelet refinement switched off: row N3.
    with_grad: also what the reference's autograd returns for a fixed upstream gradient `dout` — d/dx and d/d(every parameter that
    takes part): the training half of row N3 (one fused mix launch under autograd must reproduce them)."""
    torch.manual_seed(seed)
    mh = ref.SpectreMultiHead(E, H, n_fft, pooling_type="mean", num_groups=G, wavelet_on_rate=0.0).eval()
    g = torch.Generator().manual_seed(seed + 5000)
    x = torch.randn(B, N, E, generator=g)
    F = n_fft // 2 + 1
    d_out = {"x": x.numpy(), "n_fft": np.int64(n_fft), "G": np.int64(G), "H": np.int64(H)}
    mem = pp = None
    if with_mem:
        mem = torch.complex(torch.randn(F, E, generator=g), torch.randn(F, E, generator=g)) * 0.5
        d_out["mem"] = mem.numpy()
    if with_phase:
        pp = torch.exp(1j * 2 * np.pi * torch.arange(F, dtype=torch.float32) * 2.0 / n_fft).to(torch.complex64)
        d_out["pos_phase"] = pp.numpy()
    with torch.no_grad():
        d_out["out"] = mh(x, pos_phase=pp, memory_fft=mem).numpy()
    if with_grad:
        xg = x.clone().requires_grad_(True)
        out = mh(xg, pos_phase=pp, memory_fft=mem)
        dout = torch.randn(out.shape, generator=g)
        (out * dout).sum().backward()
        d_out["dout"] = dout.numpy()
        d_out["grad_x"] = xg.grad.numpy()
        for k, prm in mh.named_parameters():
            if prm.grad is not None:
                d_out["grad/" + k] = prm.grad.numpy()
    for k, v in mh.state_dict().items():
        d_out["sd/" + k] = v.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d_out)
    print(f"{name:28s} multihead x{tuple(x.shape)} H={H}{' + gradients' if with_grad else ''}  {os.path.getsize(path) / 1e3:.0f} kB")


def case_block(name, B, N, E, H, n_fft, G, seed, *, memory_size=0, with_grad=False, train_memory=False):
    """SpectreBlock.forward (spectre.py:892-982), the outermost caller of the path: pre-norm residual block around the multi-head mix,
    with its frozen spectral memory (`memory_size` bins, zero-padded to n_fft // 2 + 1 at :973-977 and chunked per head at :706-707);
    wavelet refinement off.  with_grad: the reference's autograd for a fixed upstream gradient; train_memory: with the memory
    un-frozen (`requires_grad_(True)`), so that d/d(memory_fft) — through the pad and the irfft — is in the fixture as well."""
    torch.manual_seed(seed)
    blk = ref.SpectreBlock(E, H, n_fft, pooling_type="mean", num_groups=G, wavelet_on_rate=0.0, memory_size=memory_size).eval()
    if train_memory:
        blk.memory_fft.requires_grad_(True)
    g = torch.Generator().manual_seed(seed + 6000)
    x = torch.randn(B, N, E, generator=g)
    d_out = {"x": x.numpy(), "n_fft": np.int64(n_fft), "G": np.int64(G), "H": np.int64(H), "memory_size": np.int64(memory_size)}
    with torch.no_grad():
        d_out["out"] = blk(x).numpy()
    if with_grad:
        xg = x.clone().requires_grad_(True)
        out = blk(xg)
        dout = torch.randn(out.shape, generator=g)
        (out * dout).sum().backward()
        d_out["dout"] = dout.numpy()
        d_out["grad_x"] = xg.grad.numpy()
        for k, prm in blk.named_parameters():
            if prm.grad is not None:
                d_out["grad/" + k] = prm.grad.numpy()
    for k, v in blk.state_dict().items():
        d_out["sd/" + k] = v.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d_out)
    print(f"{name:28s} block x{tuple(x.shape)} H={H} memory_size={memory_size}{' + gradients' if with_grad else ''}  {os.path.getsize(path) / 1e3:.0f} kB")


def _mask_the_reference_draws(seed, B, on_rate):
    """The coin flip of WaveletRefinement.forward (spectre.py:841) is the first use of the global generator in the forwards recorded below
    (dropout is off), so re-seeding and drawing the same shape reproduces it."""
    torch.manual_seed(seed)
    return (torch.rand(B, 1, 1) < on_rate).view(B)


def case_wavelet(name, B, N, d, seed, on_rate):
    """WaveletRefinement.forward (spectre.py:834-887) alone: v, q_pool -> v + (R(v).detach() * gate_mlp(q_pool)) * on_mask, the mask it drew,
    the gate it computed, and what the reference's autograd returns for a fixed upstream gradient (d/dv, d/dq_pool, d/d gate_mlp)."""
    torch.manual_seed(seed)
    wr = ref.WaveletRefinement(d, on_rate=on_rate)
    g = torch.Generator().manual_seed(seed + 7000)
    v = torch.randn(B, N, d, generator=g).requires_grad_(True)
    q = torch.randn(B, d, generator=g).requires_grad_(True)
    mask = _mask_the_reference_draws(seed + 1, B, on_rate)
    torch.manual_seed(seed + 1)
    out = wr(v, q)
    dout = torch.randn(out.shape, generator=g)
    (out * dout).sum().backward()
    d_out = {"v": v.detach().numpy(), "q_pool": q.detach().numpy(), "mask": mask.numpy(), "on_rate": np.float64(on_rate),
             "gate": wr.gate_mlp(q).detach().numpy(), "out": out.detach().numpy(), "dout": dout.numpy(),
             "grad_v": v.grad.numpy(), "grad_q_pool": q.grad.numpy() if q.grad is not None else np.zeros((B, d), np.float32)}
    for k, prm in wr.named_parameters():
        d_out["grad/" + k] = prm.grad.numpy() if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32)
    for k, val in wr.state_dict().items():
        d_out["sd/" + k] = val.numpy()
    assert mask.any() or on_rate == 0.0
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d_out)
    print(f"{name:28s} wavelet v{tuple(v.shape)} on={mask.int().tolist()}  {os.path.getsize(path) / 1e3:.0f} kB")


def case_wavelet_layer(name, kind, B, N, E, H, n_fft, G, seed, on_rate, *, memory_size=0):
    """SpectreMultiHead (kind="multihead") or SpectreBlock (kind="block") WITH the stochastic refinement (spectre.py:724): input, state_dict,
    the mask the refinement drew, output, and the reference's autograd for a fixed upstream gradient."""
    torch.manual_seed(seed)
    if kind == "multihead":
        m = ref.SpectreMultiHead(E, H, n_fft, pooling_type="mean", num_groups=G, wavelet_on_rate=on_rate).eval()
    else:
        m = ref.SpectreBlock(E, H, n_fft, pooling_type="mean", num_groups=G, wavelet_on_rate=on_rate, memory_size=memory_size).eval()
    g = torch.Generator().manual_seed(seed + 8000)
    x = torch.randn(B, N, E, generator=g)
    mask = _mask_the_reference_draws(seed + 1, B, on_rate)
    d_out = {"x": x.numpy(), "n_fft": np.int64(n_fft), "G": np.int64(G), "H": np.int64(H), "memory_size": np.int64(memory_size),
             "mask": mask.numpy(), "on_rate": np.float64(on_rate)}
    torch.manual_seed(seed + 1)
    with torch.no_grad():
        d_out["out"] = m(x).numpy()
    xg = x.clone().requires_grad_(True)
    torch.manual_seed(seed + 1)
    out = m(xg)
    assert np.array_equal(out.detach().numpy(), d_out["out"])          # same coin flips in both runs
    dout = torch.randn(out.shape, generator=g)
    (out * dout).sum().backward()
    d_out["dout"] = dout.numpy()
    d_out["grad_x"] = xg.grad.numpy()
    for k, prm in m.named_parameters():
        if prm.grad is not None:
            d_out["grad/" + k] = prm.grad.numpy()
    for k, val in m.state_dict().items():
        d_out["sd/" + k] = val.numpy()
    assert mask.any() and not mask.all()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d_out)
    print(f"{name:28s} {kind} + wavelet x{tuple(x.shape)} on={mask.int().tolist()}  {os.path.getsize(path) / 1e3:.0f} kB")


def case_toeplitz(name, B, N, d, n_fft, G, seed, bw, *, with_grad=False):
    """SpectreHead.forward WITH the Toeplitz step (spectre.py:519-521).  The reference's constructor cannot build the option on current PyTorch
    (`register_parameter` on an attribute that exists, :453-457), so the head is built without it and the parameter `_reset_parameters` (:464-474)
    would have created is attached by hand — every statement of the forward, `complex_conv1d` (:334-395) included, then runs unmodified."""
    head = _head(d, n_fft, G, seed)
    head.use_toeplitz = True
    head.toeplitz_bw = bw
    g = torch.Generator().manual_seed(seed + 9000)
    head.toeplitz_kernel = torch.nn.Parameter(torch.complex(torch.randn(2 * bw + 1, generator=g), torch.randn(2 * bw + 1, generator=g)) / math.sqrt(2 * bw + 1))
    x = torch.randn(B, N, d, generator=g)
    V, gate, out = _capture(head, x)
    _selfcheck(V, gate, None, n_fft, head.d_g, out)
    extra = {"toeplitz_bw": np.int64(bw)}
    if with_grad:
        xg = x.clone().requires_grad_(True)
        o = head(xg)
        dout = torch.randn(o.shape, generator=g)
        (o * dout).sum().backward()
        extra.update({"dout": dout.numpy(), "grad_x": xg.grad.numpy()})
        for k, prm in head.named_parameters():
            if prm.grad is not None:
                extra["grad/" + k] = prm.grad.numpy()
    _save(name, x=x, head=head, V=V, gate=gate, out=out, extra=extra)


def g_random(scale=0.3, zero_frac=0.18):
    def f(B, G, F, gen):
        z = torch.complex(torch.randn(B, G, F, generator=gen), torch.randn(B, G, F, generator=gen)) * scale
        keep = torch.rand(B, G, F, generator=gen) >= zero_frac  # modReLU zeroes ~18 % at init
        return z * keep
    return f


def g_unit(B, G, F, gen):
    return torch.ones(B, G, F, dtype=torch.complex64)


def g_single_bin(B, G, F, gen):
    z = torch.zeros(B, G, F, dtype=torch.complex64)
    for b in range(B):
        for gg in range(G):
            z[b, gg, (3 + 5 * b + 7 * gg) % F] = complex(0.5 + b, -0.25 * (gg + 1))
    return z


def g_imag_edges(B, G, F, gen):
    """Large imaginary parts at DC and Nyquist: irfft must ignore them."""
    z = g_random(0.3, 0.0)(B, G, F, gen)
    z[..., 0] = torch.complex(z[..., 0].real, torch.full_like(z[..., 0].real, 7.0))
    z[..., -1] = torch.complex(z[..., -1].real, torch.full_like(z[..., -1].real, -9.0))
    return z


def main():
    # G1 — config 0 of BASELINE.json: (B=4, N=256, D=64, G=4)
    case_module("g1_b4_n256_d64", 4, 256, 64, 256, 4, seed=0)
    # G2 — memory_fft + pos_phase in each accepted shape (spectre.py:534-536, :548-549)
    case_module("g2_mem_phaseF", 2, 64, 32, 64, 4, seed=1, with_mem=True, phase_shape="F")
    case_module("g2_mem_phase1F", 2, 64, 32, 64, 4, seed=2, with_mem=True, phase_shape="1F")
    case_module("g2_mem_phaseBF", 2, 64, 32, 64, 4, seed=3, with_mem=True, phase_shape="BF")
    case_module("g2_attnpool", 2, 64, 32, 64, 2, seed=4, pooling="attention")
    # G3 — N < n_fft (zero pad) and N > n_fft (truncate; output shrinks to n_fft rows)
    case_module("g3_pad_n20_fft32", 2, 20, 8, 32, 2, seed=5)
    case_module("g3_trunc_n48_fft32", 2, 48, 8, 32, 2, seed=6)
    # G4 — non-power-of-two even n_fft sharing N=3000's factor set (2,3,5), and 3000 itself
    case_module("g4_n60", 2, 60, 8, 60, 2, seed=7)
    case_module("g4_n24", 2, 24, 8, 24, 4, seed=8)
    case_module("g4_n3000", 1, 3000, 8, 3000, 2, seed=9, with_sd=False)
    # G5 — odd n_fft (only Im(DC) is ignored), plus a prime and a 7-smooth length
    case_module("g5_n15", 2, 15, 8, 15, 2, seed=10)
    case_fixed_gate("g5_n97_prime", 2, 97, 4, 97, 2, 11, g_random())
    case_fixed_gate("g5_n210", 1, 210, 4, 210, 1, 12, g_random())
    case_fixed_gate("g5_n31_oddG", 2, 31, 6, 31, 2, 13, g_imag_edges)   # d_g = 3 (odd): no channel pairing
    # G6 — adversarial gates through the reference's own lines :542-553
    case_fixed_gate("g6_unit_gate", 2, 64, 8, 64, 2, 14, g_unit)
    case_fixed_gate("g6_single_bin", 2, 64, 8, 64, 2, 15, g_single_bin)
    case_fixed_gate("g6_imag_edges", 2, 64, 8, 64, 4, 16, g_imag_edges)
    case_fixed_gate("g6_zeros_mem", 2, 128, 8, 128, 2, 17, g_random(0.3, 0.5), with_mem=True)
    case_fixed_gate("g6_pad_mem", 2, 50, 8, 128, 2, 18, g_random(), with_mem=True)
    # G7 — full-length columns at the benchmark sizes
    case_fixed_gate("g7_n1024", 1, 1024, 16, 1024, 4, 19, g_random())
    case_fixed_gate("g7_n4096", 1, 4096, 16, 4096, 4, 20, g_random())
    case_fixed_gate("g7_n2048_mem", 1, 2048, 16, 2048, 2, 21, g_random(), with_mem=True)
    # G9 — backward (reference autograd) at the hot path's inputs
    case_backward("g9_bwd_n64", 2, 64, 16, 64, 2, 24)
    case_backward("g9_bwd_n256", 2, 256, 32, 256, 4, 25)
    case_backward("g9_bwd_pad_n50_fft64", 2, 50, 16, 64, 2, 26)
    case_backward("g9_bwd_trunc_n80_fft64", 2, 80, 16, 64, 2, 27)
    case_backward("g9_bwd_n60", 2, 60, 12, 60, 2, 28)
    case_backward("g9_bwd_n1024", 1, 1024, 16, 1024, 2, 29)
    # G10 — prefill + autoregressive decode (ring buffer wrap, eviction, non-power-of-two and odd n_fft, memory)
    case_decode("g10_decode_n64", 32, 64, 2, 30, L=40, T=60)
    case_decode("g10_decode_n256_full", 64, 256, 4, 31, L=256, T=20)
    case_decode("g10_decode_n60", 16, 60, 2, 32, L=10, T=70)
    case_decode("g10_decode_n15_odd", 8, 15, 2, 33, L=5, T=25)
    case_decode("g10_decode_n128_mem", 32, 128, 4, 34, L=100, T=40, with_mem=True)
    # G11 — the multi-head wrapper (wavelet refinement off)
    case_multihead("g11_multihead_h2", 2, 64, 32, 2, 64, 2, 40)
    case_multihead("g11_multihead_h4_mem_phase", 2, 48, 64, 4, 64, 2, 41, with_mem=True, with_phase=True)
    case_multihead("g11_multihead_h3_grad", 2, 256, 96, 3, 256, 2, 42, with_grad=True)
    case_multihead("g11_multihead_h2_grad_mem_phase", 2, 50, 64, 2, 64, 4, 43, with_mem=True, with_phase=True, with_grad=True)
    # G12 — the transformer block: residuals, LayerNorms, MLP, and the spectral memory in its three sizes (off, all bins, truncated)
    case_block("g12_block_nomem", 2, 64, 32, 2, 64, 2, 50)
    case_block("g12_block_mem_full", 2, 48, 32, 2, 64, 2, 51, memory_size=1)
    case_block("g12_block_mem_trunc9_grad", 2, 64, 32, 2, 64, 2, 52, memory_size=9, with_grad=True)
    case_block("g12_block_mem_trunc40_trainmem", 2, 100, 48, 3, 128, 2, 53, memory_size=40, with_grad=True, train_memory=True)
    case_block("g12_block_mem_full_trainmem_odd", 2, 45, 16, 2, 45, 2, 54, memory_size=1, with_grad=True, train_memory=True)
    # G13 — the stochastic wavelet refinement (spectre.py:819-887), alone and inside the layer / the block
    case_wavelet("g13_wavelet_n64", 6, 64, 16, 60, 0.5)
    case_wavelet("g13_wavelet_n8_all_on", 2, 8, 4, 61, 1.0)
    case_wavelet("g13_wavelet_n256_d24", 4, 256, 24, 62, 0.5)
    case_wavelet("g13_wavelet_n2", 3, 2, 8, 63, 0.7)
    case_wavelet("g13_wavelet_n1024_d5", 3, 1024, 5, 64, 0.6)
    case_wavelet_layer("g13_layer_multihead_wavelet", "multihead", 4, 64, 32, 2, 64, 2, 65, 0.5)
    case_wavelet_layer("g13_layer_block_wavelet_mem", "block", 4, 128, 32, 2, 128, 2, 66, 0.5, memory_size=9)
    # G14 — the Toeplitz option of the gate producer (the reference's forward with the parameter its constructor fails to create)
    case_toeplitz("g14_toeplitz_bw4", 2, 64, 32, 64, 2, 70, 4)
    case_toeplitz("g14_toeplitz_bw2_grad", 2, 128, 16, 128, 4, 71, 2, with_grad=True)
    # G8 — bf16 input values; oracle = reference on x_bf16.float(); bf16 rounding of the result stored
    case_fixed_gate("g8_bf16_n1024", 1, 1024, 16, 1024, 4, 22, g_random(), bf16=True)
    case_fixed_gate("g8_bf16_n4096", 1, 4096, 16, 4096, 4, 23, g_random(), bf16=True)


if __name__ == "__main__":
    # optional name prefixes: `python make_golden.py g10` regenerates only the matching cases
    if len(sys.argv) > 1:
        _only = tuple(sys.argv[1:])
        for _fn in ("case_module", "case_fixed_gate", "case_backward", "case_decode", "case_multihead", "case_block", "case_wavelet", "case_wavelet_layer", "case_toeplitz"):
            def _wrap(f):
                return lambda name, *a, **k: f(name, *a, **k) if name.startswith(_only) else None
            globals()[_fn] = _wrap(globals()[_fn])
    main()
