"""Training step of one SpectreHead (forward + backward) with the gate producer tail as ONE autograd node vs the reference's ~12 ATen ops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import SpectreHead
dev = "cuda:0"
for (B, N, D) in [(64, 4096, 768), (256, 1024, 768), (16, 4096, 768)]:
    head = SpectreHead(D, N, num_groups=4, pooling_type="mean").to(dev)
    x = torch.randn(B, N, D, device=dev, requires_grad=True)
    for fused in (True, False, True, False):
        head.fused_gate_autograd = fused
        for _ in range(2):
            head(x).sum().backward()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            head(x).sum().backward()
        e1.record(); torch.cuda.synchronize()
        print(f"({B},{N},{D}) fused gate node={fused}: {e0.elapsed_time(e1) / 5:.3f} ms per forward+backward")
