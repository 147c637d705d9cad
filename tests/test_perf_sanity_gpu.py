"""Coarse performance sanity on the GPU: bf16 rows move half the bytes of fp32 rows, so a length where the bf16 launch takes much longer
than the fp32 one has a code-generation problem.  (Round 3: hipcc had serialised the bf16 loads of five mixed-radix lengths and of the
bf16 gate gradient — one request in flight per wave, 2 x slower, invisible to every parity test; tools/dtype_sweep.py,
tools/serial_load_scan.py.)  Thresholds are loose (1.4 x): this catches pathologies, not regressions of a few percent.  A comparison that fails is repeated twice before the
test fails (a pathology reproduces, a noisy neighbour does not), and conftest.py collects this module LAST so that under `-x` it cannot
hide a parity test."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.mark.parametrize("n", [1536, 1920, 2560, 3072, 3840, 1024, 4096])
def test_bf16_rows_are_not_slower_than_fp32_rows(n):
    from fft_amd import time_kernel
    dev = _dev()
    B = (96 * 3000) // n
    g = torch.randn(B, 4, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    t = {}
    for attempt in range(3):
        for dt in (torch.float32, torch.bfloat16):
            V = torch.randn(B, n, 768, device=dev).to(dt)
            out = torch.empty_like(V)
            t[dt] = min(time_kernel(V, g, None, n, out=out, warmup=8, iters=5) for _ in range(2))
        if t[torch.bfloat16] <= 1.4 * t[torch.float32]:
            break
    assert t[torch.bfloat16] <= 1.4 * t[torch.float32], (n, t)


@pytest.mark.parametrize("n", [960, 1536, 1920, 2000])
def test_bf16_gate_gradient_is_not_slower_than_fp32(n):
    from fft_amd import spectral_mix_backward
    dev = _dev()
    B = (96 * 3000) // n
    g = torch.randn(B, 4, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    t = {}
    for attempt in range(3):
        for dt in (torch.float32, torch.bfloat16):
            V = torch.randn(B, n, 768, device=dev).to(dt)
            do = torch.randn(B, n, 768, device=dev).to(dt)
            for _ in range(4):
                spectral_mix_backward(V, g, do, n, need_dv=False, need_dgate=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                spectral_mix_backward(V, g, do, n, need_dv=False, need_dgate=True)
            e1.record(); torch.cuda.synchronize()
            t[dt] = e0.elapsed_time(e1) / 4
        if t[torch.bfloat16] <= 1.4 * t[torch.float32]:
            break
    assert t[torch.bfloat16] <= 1.4 * t[torch.float32], (n, t)
