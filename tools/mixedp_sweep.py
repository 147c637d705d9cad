"""Persistent mixed-radix kernels at (256, n, 768) plus a parity check; run again with SPECTRE_MIXEDP=0 for the one-tile-per-workgroup kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd.functional import spectral_mix, describe, time_kernel
from oracle.spectral_mix_oracle import spectral_mix_numpy, assert_close
dev = "cuda:0"
for n in (3000, 3840, 3600, 3072, 2560, 2400):
    torch.manual_seed(n)
    V = torch.randn(5, n - 7, 80, device=dev); g = torch.randn(5, 5, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    y = spectral_mix(V, g, None, n); torch.cuda.synchronize()
    err = assert_close(y.cpu().numpy(), spectral_mix_numpy(V.cpu().numpy(), g.cpu().numpy(), None, n), what=str(n))
    B, D, G = 256, 768, 4
    V = torch.randn(B, n, D, device=dev); g = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    out = torch.empty_like(V)
    ms = min(time_kernel(V, g, None, n, out=out, warmup=3, iters=10) for _ in range(3))
    print(f"n={n}: {ms:.3f} ms  parity err/rms {err:.1e}  [{describe(V, g, None, n)[:44]}]")
