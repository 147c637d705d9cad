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


# ---- bf16 rows in, fp32 rows out (BASELINE.json configs[2]: "bf16 in / fp32 compute"): the same kernel with 8-byte lane accesses,
#      quads of workgroups on adjacent tiles.  The oracle is given the bf16-rounded input, so the tolerance is the fp32 one.
@pytest.mark.parametrize("B,Nin,D,G", [(1, 4096, 16, 1), (3, 4096, 64, 4), (5, 4096, 80, 5), (37, 4096, 112, 7), (2, 4096, 48, 3),
                                       (3, 4000, 64, 2), (2, 1000, 48, 3), (1, 1, 16, 1), (2, 5000, 32, 2)])
def test_bf16_in_f32_out_matches_oracle(B, Nin, D, G):
    from fft_amd import spectral_mix, describe
    torch.manual_seed(B * 13 + D + Nin)
    V = torch.randn(B, Nin, D, device=DEV).bfloat16()
    gate = (torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3)
    assert describe(V, gate, None, N, out_dtype=torch.float32).startswith("regtile-pipelined 64x64 in=bf16 out=f32")
    y = spectral_mix(V, gate, None, N, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert y.dtype == torch.float32 and y.shape == (B, min(Nin, N), D)
    ref = spectral_mix_numpy(V.float().cpu().numpy(), gate.cpu().numpy(), None, N)
    assert_close(y.cpu().numpy(), ref, what=f"bf16 in ({B},{Nin},{D})")


def test_bf16_in_many_tiles_and_guard_rows():
    """Headline width, quads spanning batch boundaries, a view with a larger row stride; rows beyond N_out stay untouched."""
    from fft_amd import spectral_mix
    torch.manual_seed(11)
    B, Nin, D, G = 21, 3900, 768, 4
    Vbig = torch.randn(B, Nin, D + 32, device=DEV).bfloat16()
    V = Vbig[:, :, 16:16 + D]                                   # 32-byte offset, row stride D + 32: 8-byte aligned lanes
    gate = (torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3)
    out = torch.full((B, Nin + 3, D), 7.0, device=DEV)
    y = spectral_mix(V, gate, None, N, out=out[:, :Nin], out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert torch.all(out[:, Nin:] == 7.0)
    d_g = D // G
    for (b, c) in [(0, 0), (B - 1, D - 2), (B // 2, 18), (7, D // 2 + 2), (B - 2, 16 * 13 + 4)]:
        ref = spectral_mix_numpy(V[b:b + 1, :, c:c + 2].float().cpu().numpy(), gate[b:b + 1, c // d_g:c // d_g + 1].cpu().numpy(), None, N)
        assert_close(y[b:b + 1, :, c:c + 2].cpu().numpy(), ref, what=f"bf16 in column ({b},{c})")


def test_bf16_in_with_memory_fft_keeps_the_round1_kernel():
    from fft_amd import describe
    V = torch.randn(2, N, 32, device=DEV).bfloat16()
    gate = torch.randn(2, 2, N // 2 + 1, dtype=torch.complex64, device=DEV)
    mem = torch.randn(N // 2 + 1, 32, dtype=torch.complex64, device=DEV)
    assert "pipelined" not in describe(V, gate, mem, N, out_dtype=torch.float32)


def test_bf16_in_unaligned_view_falls_back_and_agrees():
    """A bf16 channel slice that starts 4 bytes into a row is not 8-byte aligned: the round-1 kernel runs, same numbers (the two
    kernels order their butterflies differently: fp32 rounding level, not bit-identical)."""
    from fft_amd import describe, spectral_mix
    torch.manual_seed(3)
    Vd = torch.randn(3, N, 128, device=DEV).bfloat16()
    gd = (torch.randn(3, 4, N // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3)
    big = torch.zeros(3, N, 128 + 8, device=DEV, dtype=torch.bfloat16)
    big[:, :, 2:130] = Vd
    view = big[:, :, 2:130]
    assert describe(view, gd, None, N, out_dtype=torch.float32).startswith("regtile 64x64")
    assert describe(Vd, gd, None, N, out_dtype=torch.float32).startswith("regtile-pipelined 64x64")
    y_view = spectral_mix(view, gd, None, N, out_dtype=torch.float32)
    y = spectral_mix(Vd, gd, None, N, out_dtype=torch.float32)
    torch.cuda.synchronize()
    rms = float(y.square().mean().sqrt())
    assert float((y - y_view).abs().max()) <= 1e-4 * rms


@pytest.mark.parametrize("B,Nin,D,G", [(1, 4096, 16, 1), (3, 4096, 64, 4), (37, 4096, 112, 7), (3, 4000, 64, 2), (2, 1000, 48, 3), (2, 5000, 32, 2)])
def test_bf16_in_bf16_out_is_the_rounded_f32_result(B, Nin, D, G):
    """bf16 rows out = round-to-nearest-even of what the bf16 -> f32 variant stores (same arithmetic, same kernel), bit for bit; and that
    is within the fp32 tolerance of the oracle (checked above), so the bf16 result is the oracle's within one bf16 rounding."""
    from fft_amd import spectral_mix, describe
    torch.manual_seed(B + Nin + D)
    V = torch.randn(B, Nin, D, device=DEV).bfloat16()
    gate = (torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3)
    assert describe(V, gate, None, N).startswith("regtile-pipelined 64x64 in=bf16 out=bf16")
    yb = spectral_mix(V, gate, None, N)
    yf = spectral_mix(V, gate, None, N, out_dtype=torch.float32)
    torch.cuda.synchronize()
    assert yb.dtype == torch.bfloat16 and yb.shape == (B, min(Nin, N), D)
    assert torch.equal(yb, yf.bfloat16())
    ref = spectral_mix_numpy(V.float().cpu().numpy(), gate.cpu().numpy(), None, N)
    assert_close(yb.float().cpu().numpy(), ref, rtol=1e-2, atol_rms=1e-2, what=f"bf16 out ({B},{Nin},{D})")


# ---- n_fft = 3000 (BASELINE.json configs[4]): persistent kernel with deferred row blocks (kernel_regtile_mixedp.h) ---------------------
@pytest.mark.parametrize("B,Nin,D,G", [(1, 3000, 16, 1), (3, 3000, 64, 4), (5, 3000, 80, 5), (37, 3000, 112, 7), (2, 2900, 48, 3),
                                       (2, 700, 32, 2), (1, 1, 16, 1), (2, 3500, 32, 2)])
def test_n3000_persistent_matches_oracle(B, Nin, D, G):
    """tiles = B*D/16: 1, 12, 25 (odd: a workgroup pair with one tile missing), 259 (two tiles per workgroup, odd), padded and truncated."""
    from fft_amd import describe, spectral_mix
    n = 3000
    torch.manual_seed(B * 7 + D + Nin)
    V = torch.randn(B, Nin, D, device=DEV)
    gate = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3
    gate = gate * (torch.rand(B, G, n // 2 + 1, device=DEV) >= 0.15)
    assert describe(V, gate, None, n).startswith("regtile-mixed-pipelined 60x50")
    y = spectral_mix(V, gate, None, n)
    torch.cuda.synchronize()
    assert y.shape == (B, min(Nin, n), D)
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.cpu().numpy(), gate.cpu().numpy(), None, n), what=f"n3000 ({B},{Nin},{D})")


@pytest.mark.parametrize("n", [2560, 2400, 3072, 3600, 3840])
@pytest.mark.parametrize("B,dN,D,G", [(1, 0, 16, 1), (5, 0, 80, 5), (37, 0, 112, 7), (3, -100, 64, 2), (2, 333, 32, 2), (40, 0, 768, 4)])
def test_other_persistent_mixed_lengths_match_oracle(n, B, dN, D, G):
    from fft_amd import describe, spectral_mix
    torch.manual_seed(n + B + D)
    V = torch.randn(B, n + dN, D, device=DEV)
    gate = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3
    assert describe(V, gate, None, n).startswith("regtile-mixed-pipelined")
    y = spectral_mix(V, gate, None, n)
    torch.cuda.synchronize()
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.cpu().numpy(), gate.cpu().numpy(), None, n), what=f"n={n} ({B},{n + dN},{D})")


@pytest.mark.parametrize("n", [3000, 3600])
def test_persistent_mixed_lengths_input_views_and_alignment(n):
    """Round 4: the persistent mixed-radix kernels request 16 row blocks of the next tile as 16-byte LDS-DMA lanes.  A channel-chunk view
    (row stride 3 * D, 16-byte aligned) goes through them; a view whose first element is only 8-byte aligned takes the one-tile-per-workgroup
    kernel instead (same results either way); several tiles per workgroup so that staged, deferred and reloaded blocks all occur."""
    from fft_amd import describe, spectral_mix
    torch.manual_seed(n)
    B, D, G = 70, 96, 2
    big = torch.randn(B, n, 3 * D + 4, device=DEV)
    gate = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3
    for off, persistent in ((D, True), (D + 4, True), (D + 2, False)):
        V = big[:, :, off:off + D]
        assert (V.data_ptr() % 16 == 0) == persistent
        assert describe(V, gate, None, n).startswith("regtile-mixed-pipelined") == persistent
        y = spectral_mix(V, gate, None, n)
        torch.cuda.synchronize()
        ref = spectral_mix(V.contiguous(), gate, None, n, algo="stockham")
        rms = float(ref.square().mean().sqrt())
        assert float((y - ref).abs().max()) <= 2e-4 * rms
        for (b, c) in [(0, 0), (B - 1, D - 2), (B // 2, 18)]:
            want = spectral_mix_numpy(V[b:b + 1, :, c:c + 2].cpu().numpy(), gate[b:b + 1, c // (D // G):c // (D // G) + 1].cpu().numpy(), None, n)
            assert_close(y[b:b + 1, :, c:c + 2].cpu().numpy(), want, what=f"n={n} view offset {off} column ({b},{c})")


def test_n3000_persistent_many_tiles_guard_rows_and_repeatability():
    """Headline width with several tiles per workgroup; rows beyond N_out stay untouched; two launches are bit-identical; agreement with
    the one-tile-per-workgroup kernel it replaces (algo="stockham" is a third implementation) on whole tensors, oracle on columns."""
    from fft_amd import spectral_mix
    n = 3000
    torch.manual_seed(5)
    B, Nin, D, G = 40, 2950, 768, 4
    V = torch.randn(B, Nin, D, device=DEV)
    gate = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=DEV) * 0.3
    out = torch.full((B, Nin + 2, D), 7.0, device=DEV)
    y = spectral_mix(V, gate, None, n, out=out[:, :Nin])
    y2 = spectral_mix(V, gate, None, n)
    y3 = spectral_mix(V, gate, None, n, algo="stockham")
    torch.cuda.synchronize()
    assert torch.all(out[:, Nin:] == 7.0)
    assert torch.equal(y, y2)
    rms = float(y3.square().mean().sqrt())
    assert float((y - y3).abs().max()) <= 2e-4 * rms
    d_g = D // G
    for (b, c) in [(0, 0), (B - 1, D - 2), (B // 2, 18), (7, D // 2 + 2), (B - 2, 16 * 13 + 4)]:
        ref = spectral_mix_numpy(V[b:b + 1, :, c:c + 2].cpu().numpy(), gate[b:b + 1, c // d_g:c // d_g + 1].cpu().numpy(), None, n)
        assert_close(y[b:b + 1, :, c:c + 2].cpu().numpy(), ref, what=f"n3000 column ({b},{c})")
