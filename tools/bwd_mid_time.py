"""dV / dgate timings at the mid-size mixed-radix lengths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix_backward
dev = "cuda:0"
for (B, N, D) in [(256, 2000, 768), (256, 1920, 768), (256, 1536, 768), (256, 1200, 768)]:
    V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; do = torch.randn(B, N, D, device=dev)
    for need_dv, need_dg, name in ((True, False, "dV"), (False, True, "dgate")):
        for _ in range(2): spectral_mix_backward(V, g, do, N, need_dv=need_dv, need_dgate=need_dg)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): spectral_mix_backward(V, g, do, N, need_dv=need_dv, need_dgate=need_dg)
        torch.cuda.synchronize(); print(f"({B},{N},{D}) {name}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
