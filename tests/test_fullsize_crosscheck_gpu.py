"""Whole-tensor comparison of the deep persistent kernels (48 tiles per workgroup at the benchmark shapes) against the THIRD implementation
of the path, the LDS Stockham kernel (`algo="stockham"`: no register tiles, no pipelining, no LDS-DMA, no hand-counted waits).  A stale
gate, a wrong tile or a slot read before it landed that is consistent from launch to launch would pass the property tests of
test_parity_gpu.py and fail here.  The Stockham kernel itself is pinned by the oracle on 120 shapes (test_parity_gpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _inputs(B, N, D, G, dt, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    V = torch.randn(B, N, D, device=DEV, generator=g).to(dt)
    gate = torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device=DEV, generator=g) * 0.3
    gate = gate * (torch.rand(B, G, N // 2 + 1, device=DEV, generator=g) >= 0.18)          # exact zeros, like the modReLU output
    return V, gate


def _chunked_max_diff(a, b, chunk=32):
    worst = 0.0
    for i in range(0, a.shape[0], chunk):
        worst = max(worst, float((a[i:i + chunk].float() - b[i:i + chunk].float()).abs().max()))
    return worst


@pytest.mark.parametrize("B,N,D,G,kernel", [(256, 4096, 768, 4, "regtile-pipelined 64x64"), (256, 3000, 768, 4, "regtile-mixed-pipelined 60x50")],
                         ids=["C2_f32_256x4096x768", "C4_f32_256x3000x768"])
def test_fp32_whole_tensor_against_stockham(B, N, D, G, kernel):
    from fft_amd import describe, spectral_mix
    V, gate = _inputs(B, N, D, G, torch.float32, seed=1234 + N)
    assert describe(V, gate, None, N).startswith(kernel)
    y = spectral_mix(V, gate, None, N)
    ref = torch.empty_like(y)
    for i in range(0, B, 64):                               # the Stockham launch in batch chunks (bounds its running time per launch)
        spectral_mix(V[i:i + 64], gate[i:i + 64], None, N, out=ref[i:i + 64], algo="stockham")
    torch.cuda.synchronize()
    rms = float(ref[:8].square().mean().sqrt())
    # two fp32 implementations with different butterfly orders: each within ~1e-6 RMS of the exact result (test_parity_gpu.py)
    assert _chunked_max_diff(y, ref) <= 2e-5 * rms * 10
    # and not a single NaN / untouched element
    assert bool(torch.isfinite(y).all())


def test_bf16_out_is_the_rounding_of_the_fp32_out_variant_at_full_size():
    """(256, 4096, 768) bf16 -> bf16 (gangs of four workgroups, 8-byte lane stores): bit-equal to rounding the bf16 -> fp32 variant's
    output, which in turn is compared with Stockham on the same bf16 input."""
    from fft_amd import describe, spectral_mix
    B, N, D, G = 256, 4096, 768, 4
    V, gate = _inputs(B, N, D, G, torch.bfloat16, seed=77)
    assert describe(V, gate, None, N).startswith("regtile-pipelined 64x64 in=bf16 out=bf16")
    y16 = spectral_mix(V, gate, None, N)                                    # bf16 out (native dtype of the input)
    y32 = spectral_mix(V, gate, None, N, out_dtype=torch.float32)          # same arithmetic, fp32 rows out
    torch.cuda.synchronize()
    assert y16.dtype == torch.bfloat16 and y32.dtype == torch.float32
    for i in range(0, B, 32):
        assert torch.equal(y16[i:i + 32], y32[i:i + 32].bfloat16()), f"batch chunk {i}"
    ref = torch.empty_like(y32)
    for i in range(0, B, 64):
        spectral_mix(V[i:i + 64], gate[i:i + 64], None, N, out=ref[i:i + 64], algo="stockham")
    torch.cuda.synchronize()
    rms = float(ref[:8].square().mean().sqrt())
    assert _chunked_max_diff(y32, ref) <= 2e-5 * rms * 10


def test_full_size_properties_bf16_native_output():
    """(256, 4096, 768) bf16 in -> bf16 out, the output dtype the kernel picks for a bf16 input (BASELINE configs[2] read as bf16 I/O):
    the size-independent properties of test_parity_gpu.py::test_full_size_properties, with tolerances in bf16 ulps."""
    import numpy as np
    from fft_amd import spectral_mix
    from oracle.spectral_mix_oracle import spectral_mix_numpy
    B, N, D, G = 256, 4096, 768, 4
    F = N // 2 + 1
    V, gate = _inputs(B, N, D, G, torch.bfloat16, seed=5)

    rms = float(V[:8].float().square().mean().sqrt())

    def ulps(a, b):                      # distance in bf16 units in the last place, after an absolute floor of 4e-6 RMS (the fp32 arithmetic
        a32, b32 = a.float(), b.float()  # is good to ~1e-6 of the RMS, which is many ulps of an element that happens to be tiny)
        scale = torch.maximum(a32.abs(), b32.abs()).clamp_min(2.0 ** -120)
        ulp = torch.exp2(torch.floor(torch.log2(scale)) - 7)
        return float((((a32 - b32).abs() - 4e-6 * rms).clamp_min(0) / ulp).max())

    # (1) unit gate with junk in Im(DC) / Im(Nyquist): the round trip returns the bf16 input (the fp32 result is within 1e-6 of a bf16 number)
    ones = torch.ones(B, G, F, dtype=torch.complex64, device=DEV)
    ones[..., 0] += 3j
    ones[..., -1] -= 5j
    y = spectral_mix(V, ones, None, N)
    torch.cuda.synchronize()
    assert y.dtype == torch.bfloat16
    for i in range(0, B, 64):
        assert ulps(y[i:i + 64], V[i:i + 64]) <= 1.0
    del y
    # (2) exact scaling by 2, (3) shard / concat equality bit for bit
    y1 = spectral_mix(V, gate, None, N)
    y2 = spectral_mix((V.float() * 2).bfloat16(), gate, None, N)
    torch.cuda.synchronize()
    for i in range(0, B, 64):
        assert torch.equal(y2[i:i + 64].float(), 2 * y1[i:i + 64].float())
    h = B // 2
    assert torch.equal(torch.cat([spectral_mix(V[:h], gate[:h], None, N), spectral_mix(V[h:], gate[h:], None, N)]), y1)
    # (4) whole columns against the float64 oracle on the same bf16 inputs: the bf16 rounding of the exact result, within 1 ulp
    d_g = D // G
    idx_b = [0, B // 3, B - 1]
    for c in (0, 1, D // 2 + 1, D - 1):
        Vs = V[idx_b][:, :, c:c + 1].float().cpu().numpy()
        gsel = gate[idx_b][:, c // d_g:c // d_g + 1].cpu().numpy()
        ref = torch.from_numpy(spectral_mix_numpy(Vs, gsel, None, N).astype(np.float32)).bfloat16()
        assert ulps(y1[idx_b][:, :, c:c + 1].cpu(), ref) <= 1.0, f"column {c}"
