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


@pytest.mark.parametrize("B,N_in,D,G", [(2, 4000, 32, 2), (3, 1, 16, 1), (2, 63, 48, 3), (2, 2049, 64, 4), (5, 3999, 80, 5), (2, 5000, 32, 2), (24, 3000, 192, 4)])
def test_padded_and_truncated_sequences_on_the_pipelined_kernel(B, N_in, D, G):
    """n_fft = 4096 with N_in != n_fft: rows >= N_in are never read (the buffer loads return the zero padding of spectre.py:506) and
    never written (spectre.py:553 keeps min(N, n_fft) rows); guard rows around the output must stay untouched."""
    from fft_amd import describe, spectral_mix
    g = torch.Generator().manual_seed(N_in + D)
    V = torch.randn(B, N_in, D, generator=g)
    F = N // 2 + 1
    gate = (torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)) * 0.3).to(torch.complex64)
    Vd, gd = V.to(DEV), gate.to(DEV)
    assert describe(Vd, gd, None, N).startswith("regtile-pipelined 64x64")
    n_out = min(N_in, N)
    guard = torch.full((B, n_out + 3, D), 7.25, device=DEV)            # three sentinel rows behind every batch element's output
    out = guard[:, :n_out]
    spectral_mix(Vd, gd, None, N, out=out)
    torch.cuda.synchronize()
    assert bool((guard[:, n_out:] == 7.25).all()), "rows beyond min(N, n_fft) were written"
    assert_close(out.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, N), what=f"padded ({B},{N_in},{D})")


@pytest.mark.parametrize("B,N_in,D,G", [(1, 4096, 16, 1), (3, 4096, 64, 4), (2, 4000, 48, 3), (24, 4096, 192, 4), (5, 100, 80, 5), (2, 5000, 32, 2)])
def test_memory_fft_on_the_pipelined_kernel(B, N_in, D, G):
    """spectre.py:548-549 (mixed + memory_fft before the inverse transform), incl. padded / truncated sequences and 2 tiles per workgroup."""
    from fft_amd import describe, spectral_mix
    g = torch.Generator().manual_seed(17 * N_in + D)
    V = torch.randn(B, N_in, D, generator=g)
    F = N // 2 + 1
    gate = (torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)) * 0.3).to(torch.complex64)
    mem = (torch.complex(torch.randn(F, D, generator=g), torch.randn(F, D, generator=g)) * 0.2).to(torch.complex64)
    Vd, gd, md = V.to(DEV), gate.to(DEV), mem.to(DEV)
    assert describe(Vd, gd, md, N).startswith("regtile-pipelined 64x64 in=f32 out=f32 mode=4")
    y = spectral_mix(Vd, gd, md, N)
    torch.cuda.synchronize()
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), mem.numpy(), N), what=f"memory_fft ({B},{N_in},{D})")
