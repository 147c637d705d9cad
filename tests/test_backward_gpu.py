"""spectre_mix_bwd on a real GPU: reference-autograd fixtures, the fp64 closed form at more shapes, and autograd
through the drop-in module."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.spectral_mix_oracle import assert_close, spectral_mix_backward_numpy
from test_backward_cpu import BWD, IDS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bwd(V, gate, dout, n_fft, **k):
    from fft_amd import spectral_mix_backward
    dv, dg = spectral_mix_backward(V.to(DEV), gate.to(DEV), dout.to(DEV), n_fft, **k)
    torch.cuda.synchronize()
    return dv, dg


def _check(dv, dg, dV_ref, dg_ref, what):
    assert_close(dv.float().cpu().numpy(), dV_ref, what=what + " dV")
    assert_close(torch.view_as_real(dg).cpu().numpy(), np.ascontiguousarray(dg_ref).astype(np.complex64).view(np.float32).reshape(*dg_ref.shape, 2),
                 what=what + " dgate")


@pytest.mark.parametrize("path", BWD, ids=IDS)
def test_reference_autograd_fixtures(path):
    d = load_golden(path)
    dv, dg = _bwd(torch.from_numpy(d["V"]), torch.from_numpy(d["gate"]), torch.from_numpy(d["dout"]), int(d["n_fft"]))
    _check(dv, dg, d["dV"], d["dgate"], path)


SHAPES = [(2, 4096, 64, 4, 4096), (2, 2048, 32, 2, 2048), (3, 1024, 48, 3, 1024), (2, 512, 32, 4, 512), (2, 256, 32, 2, 256),
          (2, 3000, 32, 2, 3000), (2, 97, 12, 2, 97), (2, 1000, 32, 2, 1024), (2, 5000, 32, 2, 4096), (2, 60, 6, 2, 60), (2, 64, 10, 2, 64),
          # register-tile gate gradient: ragged channel tiles (d_g = 6, 5, 20), many tiles per group (d_g = 200), short input
          (2, 256, 12, 2, 256), (2, 512, 15, 3, 512), (3, 1024, 40, 2, 1024), (1, 2048, 200, 1, 2048), (2, 100, 24, 1, 256),
          # mixed-radix register-tile gate gradient (RS even; 1000 runs it as 25 x 40)
          (2, 3000, 64, 4, 3000), (2, 2500, 24, 2, 3000), (2, 768, 32, 2, 768), (2, 1536, 40, 2, 1536), (1, 3072, 32, 2, 3072),
          (2, 2000, 32, 4, 2000), (2, 1280, 32, 2, 1280), (1, 2560, 48, 2, 2560), (1, 3840, 32, 2, 3840), (2, 1000, 32, 2, 1000),
          (3, 64, 32, 2, 64), (2, 128, 24, 2, 128), (2, 196, 32, 2, 196), (2, 384, 32, 2, 384), (2, 640, 32, 2, 640), (2, 960, 20, 2, 960),
          (2, 1200, 32, 2, 1200), (1, 1920, 32, 2, 1920), (1, 2400, 32, 2, 2400), (1, 3600, 32, 2, 3600), (2, 150, 32, 2, 196),
          # lane-pair gate gradient (n_fft = RF x 128), ragged channel tiles, short input
          (2, 8192, 32, 2, 8192), (1, 8192, 24, 4, 8192), (1, 6000, 20, 2, 8192), (1, 6144, 16, 1, 6144), (2, 6100, 12, 2, 6144),
          # two-pass gate gradient (spectra of V and dOut, then a reduction): every length the forward accepts has a backward
          (2, 16384, 16, 2, 16384), (1, 12288, 24, 3, 12288), (2, 15000, 12, 2, 16384), (1, 20000, 8, 1, 16384), (1, 12288, 10, 2, 12288)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"B{s[0]}_N{s[1]}_D{s[2]}_G{s[3]}_fft{s[4]}" for s in SHAPES])
def test_random_vs_fp64_closed_form(shape):
    B, N, D, G, n_fft = shape
    g = torch.Generator().manual_seed(sum(shape))
    V = torch.randn(B, N, D, generator=g)
    F = n_fft // 2 + 1
    gate = (torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)) * 0.3).to(torch.complex64)
    dout = torch.randn(B, min(N, n_fft), D, generator=g)
    dV_ref, dg_ref = spectral_mix_backward_numpy(V.numpy(), gate.numpy(), dout.numpy(), n_fft)
    dv, dg = _bwd(V, gate, dout, n_fft)
    _check(dv, dg, dV_ref, dg_ref, str(shape))
    # each output alone
    dv2, none = _bwd(V, gate, dout, n_fft, need_dgate=False)
    assert none is None and torch.equal(dv2, dv)
    none, dg2 = _bwd(V, gate, dout, n_fft, need_dv=False)
    assert none is None
    assert_close(torch.view_as_real(dg2).cpu().numpy(), torch.view_as_real(dg).cpu().numpy(), rtol=1e-5, atol_rms=1e-5)   # Stockham path: atomics, order varies


@pytest.mark.parametrize("n", [1024, 3000, 196, 4096, 300])
def test_bf16_backward(n):
    g = torch.Generator().manual_seed(5)
    F = n // 2 + 1
    V = torch.randn(2, n, 32, generator=g).bfloat16()
    gate = (torch.complex(torch.randn(2, 2, F, generator=g), torch.randn(2, 2, F, generator=g)) * 0.3).to(torch.complex64)
    dout = torch.randn(2, n, 32, generator=g).bfloat16()
    dV_ref, dg_ref = spectral_mix_backward_numpy(V.float().numpy(), gate.numpy(), dout.float().numpy(), n)
    dv, dg = _bwd(V, gate, dout, n)
    assert dv.dtype == torch.bfloat16
    assert_close(dv.float().cpu().numpy(), dV_ref, rtol=1e-2, atol_rms=1e-2, what="bf16 dV")      # bf16 storage of dV
    assert_close(torch.view_as_real(dg).cpu().numpy(), dg_ref.astype(np.complex64).view(np.float32).reshape(2, 2, F, 2), what="dgate")


def test_module_autograd_matches_torch_fft_autograd():
    """Gradients of the drop-in module's parameters == gradients of the same module with the hot path written in
    torch.fft (the reference's statements), both on the GPU."""
    from fft_amd import SpectreHead
    torch.manual_seed(0)
    head = SpectreHead(32, 256, num_groups=2, pooling_type="mean").to(DEV)
    x = torch.randn(2, 256, 32, device=DEV)
    w = torch.randn(2, 256, 32, device=DEV)
    (head(x) * w).sum().backward()
    got = {k: p.grad.clone() for k, p in head.named_parameters()}
    head.zero_grad()
    V, gate, _ = head.spectral_gate(x)
    vf = torch.fft.rfft(V, n=256, dim=1)
    y = torch.fft.irfft(gate.permute(0, 2, 1).repeat_interleave(head.d_g, dim=-1) * vf, n=256, dim=1)[:, :256]
    (y * w).sum().backward()
    for k, p in head.named_parameters():
        ref = p.grad
        scale = float(ref.abs().max()) + 1e-12
        assert float((got[k] - ref).abs().max()) <= 2e-3 * scale, k


# ------------------------------------------------------------------------------------------------------
# full benchmark sizes (BASELINE.json configs C2 and C4): columns of dV and whole (b, g) rows of dgate against the float64
# closed form — the S-way partial-sum path of the gate gradient only grows to its full width at these sizes
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,D,G", [(256, 4096, 768, 4), (256, 3000, 768, 4)])
def test_full_size_backward_columns_and_gate_rows(B, N, D, G):
    from fft_amd import spectral_mix_backward
    torch.manual_seed(1)
    F = N // 2 + 1
    V = torch.randn(B, N, D, device=DEV)
    dY = torch.randn(B, N, D, device=DEV)
    gate = torch.randn(B, G, F, dtype=torch.complex64, device=DEV) * 0.3
    gate = gate * (torch.rand(B, G, F, device=DEV) >= 0.18)
    dV, dG = spectral_mix_backward(V, gate, dY, N)
    torch.cuda.synchronize()
    d_g = D // G
    for (b, c) in [(0, 0), (B - 1, D - 2), (B // 3, D // 2 + 2), (7, 16 * 11 + 6)]:
        grp = c // d_g
        rV, _ = spectral_mix_backward_numpy(V[b:b + 1, :, c:c + 2].cpu().numpy(), gate[b:b + 1, grp:grp + 1].cpu().numpy(),
                                            dY[b:b + 1, :, c:c + 2].cpu().numpy(), N)
        assert_close(dV[b:b + 1, :, c:c + 2].cpu().numpy(), rV, what=f"dV column ({b},{c})")
    for (b, g) in [(0, 0), (B - 1, G - 1), (B // 2, 1)]:
        sl = slice(g * d_g, (g + 1) * d_g)
        _, rG = spectral_mix_backward_numpy(V[b:b + 1, :, sl].cpu().numpy(), gate[b:b + 1, g:g + 1].cpu().numpy(),
                                            dY[b:b + 1, :, sl].cpu().numpy(), N)
        got = torch.view_as_real(dG[b:b + 1, g:g + 1]).cpu().numpy()
        assert_close(got, np.stack([rG.real, rG.imag], -1), what=f"dgate row ({b},{g})")
