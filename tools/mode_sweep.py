"""Fast mode vs general modes of the register-tile kernels (row predicates, gate from global memory, memory_fft)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev = "cuda:0"
for (B, N, D, n_fft, mem) in [(256, 4096, 768, 4096, False), (256, 4000, 768, 4096, False), (256, 4096, 768, 4096, True), (256, 4096, 776, 4096, False),
                              (256, 3000, 768, 3000, False), (256, 2900, 768, 3000, False), (256, 3000, 768, 3000, True),
                              (256, 1024, 768, 1024, False), (256, 1000, 768, 1024, False), (256, 1024, 768, 1024, True)]:
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev)
    g = torch.randn(B, 4, n_fft // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    m = torch.randn(n_fft // 2 + 1, D, dtype=torch.complex64, device=dev) if mem else None
    out = torch.empty(B, min(N, n_fft), D, device=dev)
    ms = min(time_kernel(V, g, m, n_fft, out=out, warmup=2, iters=6) for _ in range(3))
    byt = B * N * D * 4 + out.numel() * 4
    print(f"(B={B}, N={N}, D={D}, n_fft={n_fft}, mem={mem}): {ms:7.3f} ms  {byt/ms/1e6:6.0f} GB/s  [{describe(V, g, m, n_fft)[:48]}]")
