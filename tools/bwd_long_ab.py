"""dgate at the lane-pair lengths (A/B between two builds: SPECTRE_HIP_LIB=<path>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix_backward
dev = "cuda:0"
for (B, N, D) in [(64, 8192, 768), (64, 6144, 768), (64, 6000, 768)]:
    n_fft = 8192 if N > 6144 else 6144
    V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, n_fft // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; do = torch.randn(B, min(N, n_fft), D, device=dev)
    for _ in range(3): spectral_mix_backward(V, g, do, n_fft, need_dv=False, need_dgate=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): spectral_mix_backward(V, g, do, n_fft, need_dv=False, need_dgate=True)
    e1.record(); torch.cuda.synchronize()
    print(f"({B},{N},{D}) n_fft={n_fft} dgate: {e0.elapsed_time(e1)/10:.3f} ms  [{os.path.basename(os.environ.get('SPECTRE_HIP_LIB', 'default'))}]")
