"""kernel_regtile_wide.h — 32-channel (whole-line) tiles at n_fft = 256 / 512 / 1024 (round 4).  Same statements of the reference
(spectre.py:506, :542-553), new geometry: parity against the fp64 oracle through the C ABI, the conditions under which the dispatcher
picks it (and leaves it), bf16 storage, conj(gate) (the dV pass of the backward), full-size properties at BASELINE config 1."""
import numpy as np
import pytest
import torch

from oracle.spectral_mix_oracle import assert_close, bf16_round, spectral_mix_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(seed, B, N, D, G, dtype=torch.float32, zero_frac=0.18):
    g = torch.Generator(device="cpu").manual_seed(seed)
    V = torch.randn(B, N, D, generator=g).to(dtype)
    F = N // 2 + 1
    gate = torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)) * 0.3
    gate = gate * (torch.rand(B, G, F, generator=g) >= zero_frac)
    return V, gate.to(torch.complex64)


def _run(V, gate, n, **kw):
    from fft_amd import spectral_mix
    y = spectral_mix(V.to(DEV), gate.to(DEV), None, n, **kw)
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("n,tag", [(256, "16x16"), (512, "32x16"), (1024, "32x32")])
@pytest.mark.parametrize("B,D,G", [(1, 32, 1), (3, 64, 2), (2, 96, 3), (5, 768, 4), (2, 768, 24)])
def test_wide_tiles_match_the_fp64_oracle(n, tag, B, D, G):
    from fft_amd import describe
    V, gate = _problem(100 + n + D + G, B, n, D, G)
    assert describe(V.to(DEV), gate.to(DEV), None, n).startswith(f"regtile-wide {tag} in=f32 out=f32 mode=0 tiles={B * D // 32}")
    y = _run(V, gate, n)
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, n), what=f"wide {n} ({B},{D},{G})")
    # the 16-channel kernel on the same problem (algo-independent answer; the two share no geometry)
    y16 = _run(V, gate, n, algo="stockham")
    assert_close(y.cpu().numpy(), y16.cpu().numpy(), what="wide vs stockham")


@pytest.mark.parametrize("n", [256, 512, 1024])
def test_wide_tiles_with_bf16_rows(n):
    from fft_amd import describe
    V, gate = _problem(7 + n, 3, n, 64, 2, dtype=torch.bfloat16)
    assert describe(V.to(DEV), gate.to(DEV), None, n).startswith("regtile-wide") and "in=bf16 out=bf16" in describe(V.to(DEV), gate.to(DEV), None, n)
    e = spectral_mix_numpy(V.float().numpy(), gate.numpy(), None, n)
    yf = _run(V, gate, n, out_dtype=torch.float32)                         # bf16 rows in, fp32 rows out
    assert yf.dtype == torch.float32
    assert_close(yf.cpu().numpy(), e, what=f"wide bf16->f32 {n}")
    yb = _run(V, gate, n)                                                  # bf16 rows in and out
    assert yb.dtype == torch.bfloat16
    assert torch.equal(yb.cpu(), yf.cpu().to(torch.bfloat16))              # RNE of the kernel's own fp32 result, bit for bit
    eb = bf16_round(e)
    d = np.abs(yb.float().cpu().numpy() - eb)
    assert np.all(d <= np.maximum(np.abs(eb) * 2.0 ** -7, 1e-4 * np.sqrt(np.mean(e ** 2)))), "more than one bf16 ulp from the oracle's rounding"


def test_dispatcher_leaves_the_wide_kernel_when_it_must():
    from fft_amd import describe
    V, gate = _problem(1, 2, 1024, 64, 2)
    Vd, gd = V.to(DEV), gate.to(DEV)
    assert describe(Vd, gd, None, 1024).startswith("regtile-wide 32x32")
    assert describe(Vd[:, :1000], gd, None, 1024).startswith("regtile-wide 32x32 in=f32 out=f32 mode=3")   # padded sequence: same kernel, buffer ranges
    V48, g48 = _problem(2, 2, 1024, 48, 1)                                  # D % 32 != 0
    assert describe(V48.to(DEV), g48.to(DEV), None, 1024).startswith("regtile 32x32")
    V2, g2 = _problem(3, 2, 1024, 64, 4)                                    # d_g = 16: a 32-channel tile would straddle two groups
    assert describe(V2.to(DEV), g2.to(DEV), None, 1024).startswith("regtile 32x32")
    F = 513
    mem = torch.complex(torch.randn(F, 64), torch.randn(F, 64)).to(torch.complex64).to(DEV)
    assert describe(Vd, gd, mem, 1024).startswith("regtile 32x32")          # memory_fft
    Vv = torch.randn(2, 1024, 128, device=DEV)[:, :, 32:96]                 # a channel-chunk view (spectre.py:703): row stride 128, 64 channels
    assert describe(Vv, gd, None, 1024).startswith("regtile-wide 32x32")
    y = _run(Vv.contiguous().cpu(), gate, 1024)
    from fft_amd import spectral_mix
    yv = spectral_mix(Vv, gd, None, 1024)
    torch.cuda.synchronize()
    assert torch.equal(y, yv)                                               # the view and its contiguous copy: bit-identical results


def test_conj_gate_pass_of_the_backward_runs_on_wide_tiles():
    """dV = mix(dOut, conj(gate)): the same kernel with the conj flag — against autograd through the oracle's torch.fft restatement."""
    from fft_amd import spectral_mix_backward
    from oracle.spectral_mix_oracle import spectral_mix_torch
    V, gate = _problem(11, 2, 1024, 64, 2)
    dout = torch.randn(2, 1024, 64, generator=torch.Generator().manual_seed(5))
    Vr = V.clone().requires_grad_(True)
    spectral_mix_torch(Vr, gate, None, 1024).backward(dout)
    dv, _ = spectral_mix_backward(V.to(DEV), gate.to(DEV), dout.to(DEV), 1024, need_dv=True, need_dgate=False)
    torch.cuda.synchronize()
    assert_close(dv.cpu().numpy(), Vr.grad.numpy(), what="dV on wide tiles")


def test_full_size_properties_at_baseline_config_1():
    """(256, 1024, 768) fp32: unit gate = identity, linearity, circular-shift equivariance, batch shard = concat, repeatability."""
    from fft_amd import describe
    B, N, D, G = 256, 1024, 768, 4
    g = torch.Generator(device=DEV).manual_seed(3)
    V = torch.randn(B, N, D, device=DEV, generator=g)
    F = N // 2 + 1
    gate = (torch.randn(B, G, F, device=DEV, generator=g) + 1j * torch.randn(B, G, F, device=DEV, generator=g)).to(torch.complex64) * 0.3
    assert describe(V, gate, None, N).startswith("regtile-wide 32x32 in=f32 out=f32 mode=0 tiles=6144")
    from fft_amd import spectral_mix
    one = torch.ones(B, G, F, dtype=torch.complex64, device=DEV)
    y1 = spectral_mix(V, one, None, N)
    rms = float(V.square().mean().sqrt())
    assert float((y1 - V).abs().max()) <= 2e-5 * rms * 10                   # identity up to fp32 round-off of a 1024-point round trip
    y = spectral_mix(V, gate, None, N)
    y2 = spectral_mix(V, gate, None, N)
    assert torch.equal(y, y2)                                               # repeatable, bit for bit
    W = torch.randn(B, N, D, device=DEV, generator=g)
    lin = spectral_mix(2.0 * V - 0.5 * W, gate, None, N) - (2.0 * y - 0.5 * spectral_mix(W, gate, None, N))
    assert float(lin.abs().max()) <= 1e-4 * float(y.square().mean().sqrt()) * 10
    sh = spectral_mix(torch.roll(V, 37, dims=1), gate, None, N)
    assert float((sh - torch.roll(y, 37, dims=1)).abs().max()) <= 1e-4 * float(y.square().mean().sqrt()) * 10
    half = torch.cat([spectral_mix(V[:128], gate[:128], None, N), spectral_mix(V[128:], gate[128:], None, N)], dim=0)
    assert torch.equal(half, y)                                             # batch shard: bit-equal
    # sampled columns against the fp64 oracle
    cols = [0, 31, 32, 383, 767]
    for b in (0, 255):
        e = spectral_mix_numpy(V[b:b + 1].cpu().numpy(), gate[b:b + 1].cpu().numpy(), None, N)
        assert_close(y[b:b + 1, :, cols].cpu().numpy(), e[:, :, cols], what=f"full-size columns, b={b}")


@pytest.mark.parametrize("n", [256, 512, 1024])
@pytest.mark.parametrize("short", [24, 1, "half"])
def test_padded_sequences_on_wide_tiles(n, short):
    """N_in < n_fft: rfft zero-pads (spectre.py:506), the output keeps N_in rows (:553) — the out-of-range case of the buffer instructions."""
    from fft_amd import describe
    N_in = n // 2 + 3 if short == "half" else n - short
    g = torch.Generator().manual_seed(n + N_in)
    V = torch.randn(3, N_in, 64, generator=g)
    F = n // 2 + 1
    gate = (torch.complex(torch.randn(3, 2, F, generator=g), torch.randn(3, 2, F, generator=g)) * 0.3).to(torch.complex64)
    assert describe(V.to(DEV), gate.to(DEV), None, n).startswith("regtile-wide") and "mode=3" in describe(V.to(DEV), gate.to(DEV), None, n)
    guard = torch.full((3, N_in + 8, 64), 7.0, device=DEV)                  # rows behind the output must stay untouched
    out = guard[:, :N_in]
    from fft_amd import spectral_mix
    y = spectral_mix(V.to(DEV), gate.to(DEV), None, n, out=out)
    torch.cuda.synchronize()
    assert y.shape == (3, N_in, 64)
    assert_close(y.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, n), what=f"padded wide {n} <- {N_in}")
    assert bool((guard[:, N_in:] == 7.0).all())
    Vb = V.to(torch.bfloat16)
    yb = _run(Vb, gate, n, out_dtype=torch.float32)
    assert_close(yb.cpu().numpy(), spectral_mix_numpy(Vb.float().numpy(), gate.numpy(), None, n), what=f"padded wide bf16 {n} <- {N_in}")


def test_wide_tiles_write_into_strided_output_views():
    """`out` as a channel-chunk view of a wider buffer (row stride 160, 64 channels at offset 32): only the view's elements change."""
    from fft_amd import spectral_mix
    V, gate = _problem(21, 3, 512, 64, 2)
    big = torch.full((3, 512, 160), -3.0, device=DEV)
    out = big[:, :, 32:96]
    spectral_mix(V.to(DEV), gate.to(DEV), None, 512, out=out)
    torch.cuda.synchronize()
    assert_close(out.cpu().numpy(), spectral_mix_numpy(V.numpy(), gate.numpy(), None, 512), what="strided out")
    assert bool((big[:, :, :32] == -3.0).all()) and bool((big[:, :, 96:] == -3.0).all())
