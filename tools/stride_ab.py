"""Row stride of `out` (and of V) under the CURRENT 4096 kernel and both tile orders: the same launch on views of wider buffers.  `out` is the
module's own allocation, so its row stride is a free choice on our side of the boundary; V's is the caller's (the W_v GEMM's output).
python tools/stride_ab.py [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import describe, set_tile_order, time_kernel

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = "cuda:0"
B, N, D, G = 256, 4096, 768, 4
F = N // 2 + 1
torch.manual_seed(0)
gate = torch.randn(B, G, F, dtype=torch.complex64, device=dev) * 0.3
Vc = torch.randn(B, N, D, device=dev)
pads = [768, 800, 832, 864, 896, 928, 960, 992, 1024, 1056, 1088, 1152, 1280]
for which in ("out", "both"):
    for order in ("tickets", "static"):
        set_tile_order(N, order)
        line = []
        for Dp in pads:
            outb = torch.empty(B, N, Dp, device=dev)
            out = outb[:, :, :D]
            if which == "both":
                vb = torch.empty(B, N, Dp, device=dev)
                vb[:, :, :D] = Vc
                V = vb[:, :, :D]
            else:
                V = Vc
            time_kernel(V, gate, None, N, out=out, warmup=30, iters=5)
            ts = sorted(time_kernel(V, gate, None, N, out=out, warmup=4, iters=12) for _ in range(rounds))
            line.append(f"{Dp * 4}B {ts[len(ts) // 2]:.4f}")
            del outb, out
        print(f"stride of {which:4s} | {order:7s} | " + "  ".join(line), flush=True)
set_tile_order(N, "auto")
print(describe(Vc, gate, None, N))
