"""Quick GPU parity of the persistent mixed-radix lengths against torch.fft on the device (fp32): tiny and multi-tile shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd.functional import spectral_mix
dev = "cuda:0"
for n in [2400, 2560, 3000, 3072, 3600, 3840]:
    for (B, D, G) in [(1, 32, 2), (2, 64, 2), (3, 768, 4), (40, 768, 4)]:
        torch.manual_seed(n + B)
        V = torch.randn(B, n, D, device=dev); g = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
        y = spectral_mix(V, g, None, n)
        gb = g.repeat_interleave(D // G, dim=1).transpose(1, 2)
        ref = torch.fft.irfft(torch.fft.rfft(V, n, dim=1) * gb, n, dim=1)
        err = (y - ref).abs().max().item(); rms = ref.pow(2).mean().sqrt().item()
        print(f"n={n} B={B} D={D}: max err {err:.3e} (rms {rms:.3e}) {'OK' if err < 1e-4 * max(1.0, ref.abs().max().item()) + 1e-4 * rms * 10 else 'FAIL'}", flush=True)
