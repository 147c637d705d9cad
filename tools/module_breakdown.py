"""Where a SpectreHead forward spends its time on the GPU (GEMMs / pooling / gate producer / fused mix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import SpectreHead, spectral_gate_fused, spectral_mix
from fft_amd.spectre import resample_complex
dev = "cuda:0"
B, N, D = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,4096,768").split(","))
head = SpectreHead(D, N, num_groups=4, pooling_type="mean").to(dev).eval()
x = torch.randn(B, N, D, device=dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    Q = head.W_q(x); V = head.W_v(x)
    qp = head.q_norm(head.pooling(Q))
    anchors = torch.view_as_complex(head.gate_mlp(qp).view(B, head.G, head.B, 2))
    def tail_ops():
        g = resample_complex(anchors, head.F_half)
        return head.modrelu(g.reshape(B, -1)).view_as(g)
    gate = tail_ops()
    print(f"B={B} N={N} D={D}")
    print(f"W_q + W_v GEMMs (fp32)              : {t(lambda: (head.W_q(x), head.W_v(x))):8.3f} ms")
    print(f"mean pooling of Q                    : {t(lambda: head.pooling(Q)):8.3f} ms")
    print(f"LN + gate MLP                        : {t(lambda: head.gate_mlp(head.q_norm(qp))):8.3f} ms")
    print(f"resample + modReLU, PyTorch ops      : {t(tail_ops):8.3f} ms")
    print(f"resample + modReLU, fused HIP (N2)   : {t(lambda: spectral_gate_fused(anchors, head.modrelu.bias, 1e-4, head.F_half)):8.3f} ms")
    print(f"fused spectral mix (HIP)             : {t(lambda: spectral_mix(V, gate, None, N)):8.3f} ms")
    print(f"whole forward                        : {t(lambda: head(x)):8.3f} ms")
