"""A few launches of every secondary kernel (backward, gate producer, prefill, decode) for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import PrefixFFTCache, SpectreHead, spectral_gate_fused, spectral_mix_backward
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, N, D) in [(256, 4096, 768), (256, 3000, 768), (256, 1024, 768)]:
    V = torch.randn(B, N, D, device=dev); do = torch.randn(B, N, D, device=dev)
    g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    for _ in range(3):
        spectral_mix_backward(V, g, do, N)
    del V, do, g
anch = torch.randn(256, 4, 45, dtype=torch.complex64, device=dev)
bias = torch.zeros(4 * 2049, device=dev)
for _ in range(3):
    spectral_gate_fused(anch, bias, 1e-4, 2049)
head = SpectreHead(768, 4096, num_groups=4, pooling_type="mean").to(dev).eval()
cache = PrefixFFTCache(4096, 768, device=dev)
cache.prefill(torch.randn(4090, 768, device=dev), torch.randn(4090, 768, device=dev))
for i in range(12):
    head.decode_step(torch.randn(768, device=dev), torch.randn(768, device=dev), cache)
torch.cuda.synchronize()
