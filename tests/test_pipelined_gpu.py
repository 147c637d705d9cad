"""Parity of the persistent, software-pipelined n_fft = 4096 kernel (fft_amd/csrc/kernel_regtile64p.h) through the C ABI:
several tiles per workgroup, odd tile counts (a workgroup pair with one tile missing), one tile, the conj-gate path of the
backward, agreement with the row-predicated kernel it replaces in the fast mode, and the fall-back for unaligned views.
Oracle: numpy float64 restatement of /root/reference/spectre.py:506,:542-553 (oracle/spectral_mix_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle.spectral_mix_oracle import assert_close, spectral_mix_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 4096


def _problem(seed, B, D, G):
    g = torch.Generator().manual_seed(seed)
    V = torch.randn(B, N, D, generator=g)
    F = N // 2 + 1
    gate = torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)) * 0.3
    gate = gate * (torch.rand(B, G, F, generator=g) >= 0.15)       # modReLU leaves exact zeros
    return V, gate.to(torch.complex64)


@pytest.mark.parametrize("B,D,G", [(1, 16, 1), (1, 32, 2), (3, 64, 4), (5, 80, 5), (24, 192, 4), (33, 208, 13), (9, 464, 29)])
def test_pipelined_matches_oracle(B, D, G):
    """tiles = B*D/16: 1, 2, 12, 25 (odd), 288 (2 tiles per workgroup), 429 (odd, 2 per workgroup), 261 (odd, 2 per workgroup)."""
    from fft_amd import describe, spectral_mix
    V, gate = _problem(B * 1000 + D, B, D, G)
    Vd, gd = V.to(DEV), gate.to(DEV)
    assert describe(Vd, gd, None, N).startswith("regtile-pipelined 64x64")
    y = spectral_mix(Vd, gd, None, N)
    torch.cuda.synchronize()
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, N), what=f"pipelined ({B},{N},{D})")


def test_pipelined_many_tiles_per_workgroup_columns():
    """(40, 4096, 768): 1920 tiles, 8 per workgroup; whole-tensor check against the kernel it replaces (bit-level agreement is
    not required: different butterfly order), column check against the fp64 oracle."""
    from fft_amd import spectral_mix
    B, D, G = 40, 768, 4
    V, gate = _problem(7, B, D, G)
    Vd, gd = V.to(DEV), gate.to(DEV)
    y = spectral_mix(Vd, gd, None, N)
    y_ref = spectral_mix(Vd, gd, None, N, algo="stockham")
    torch.cuda.synchronize()
    rms = float(y_ref.square().mean().sqrt())
    assert float((y - y_ref).abs().max()) <= 2e-5 * rms * 10
    for (b, c) in [(0, 0), (B - 1, D - 2), (17, 382), (3, 16 * 13 + 4), (B - 2, 16 * 47 + 14)]:
        grp = c // (D // G)
        ref = spectral_mix_numpy(V[b:b + 1, :, c:c + 2].numpy(), gate[b:b + 1, grp:grp + 1].numpy(), None, N)
        assert_close(y[b:b + 1, :, c:c + 2].cpu().numpy(), ref, what=f"column ({b},{c})")


def test_pipelined_in_place_safe_and_repeatable():
    """Two launches give bit-identical results (no dependence on the landing slots' previous contents or on dispatch order)."""
    from fft_amd import spectral_mix
    V, gate = _problem(11, 24, 192, 4)
    Vd, gd = V.to(DEV), gate.to(DEV)
    y1 = spectral_mix(Vd, gd, None, N).clone()
    y2 = spectral_mix(Vd, gd, None, N)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)


def test_unaligned_views_fall_back_and_agree():
    """A channel slice starting at an odd multiple of 2 floats is only 8-byte aligned: the row-predicated kernel runs, same numbers."""
    from fft_amd import describe, spectral_mix
    V, gate = _problem(5, 3, 96, 4)
    Vd, gd = V.to(DEV), gate.to(DEV)
    big = torch.zeros(3, N, 96 + 8, device=DEV)
    big[:, :, 2:98] = Vd
    view = big[:, :, 2:98]
    assert describe(view, gd, None, N).startswith("regtile 64x64")
    y_view = spectral_mix(view, gd, None, N)
    y = spectral_mix(Vd, gd, None, N)
    torch.cuda.synchronize()
    rms = float(y.square().mean().sqrt())
    assert float((y - y_view).abs().max()) <= 1e-4 * rms
    assert_close(y_view.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, N), what="unaligned view")


def test_backward_through_pipelined_kernel():
    """dV = mix(dOut, conj(gate)) runs on the pipelined kernel too (conj_gate flag); closed form in float64."""
    from fft_amd.functional import spectral_mix_backward
    from oracle.spectral_mix_oracle import spectral_mix_backward_numpy
    V, gate = _problem(3, 5, 80, 5)
    g = torch.Generator().manual_seed(99)
    dY = torch.randn(5, N, 80, generator=g)
    dV, dG = spectral_mix_backward(V.to(DEV), gate.to(DEV), dY.to(DEV), N)
    torch.cuda.synchronize()
    rV, rG = spectral_mix_backward_numpy(V.numpy(), gate.numpy(), dY.numpy(), N)
    assert_close(dV.cpu().numpy(), rV, what="dV")
    assert_close(torch.view_as_real(dG).cpu().numpy(), np.stack([rG.real, rG.imag], -1), what="dgate")
