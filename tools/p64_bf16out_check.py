"""Development aid: pipelined 4096 kernel with bf16 rows in AND out : bit-identical to torch's bf16 rounding of
the bf16 -> f32 variant's result; timing next to the round-1 kernel (SPECTRE_P64_BF16=0 in a second run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd.functional import spectral_mix, describe, time_kernel
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
for (B, Nin, D, G) in [(1, 4096, 16, 1), (3, 4096, 64, 4), (37, 4096, 112, 7), (3, 4000, 64, 2), (2, 1000, 48, 3), (2, 5000, 32, 2), (40, 4096, 768, 4)]:
    N = 4096
    V = torch.randn(B, Nin, D, device=dev).bfloat16()
    g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    yb = spectral_mix(V, g, None, N)
    yf = spectral_mix(V, g, None, N, out_dtype=torch.float32)
    torch.cuda.synchronize()
    same = torch.equal(yb, yf.bfloat16())
    ok &= same and yb.dtype == torch.bfloat16
    print(("OK  " if same else "FAIL"), (B, Nin, D, G), describe(V, g, None, N))
print("PARITY", "OK" if ok else "FAILED")
B, N, D, G = 256, 4096, 768, 4
V = torch.randn(B, N, D, device=dev).bfloat16()
g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
out = torch.empty_like(V)
byt = B * N * D * 4 + B * G * (N // 2 + 1) * 8
for rep in range(3):
    ms = time_kernel(V, g, None, N, out=out, warmup=3, iters=10)
    print(f"TIME (256,4096,768) bf16->bf16: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  frac={byt/ms/1e6/8000:.3f} [{describe(V, g, None, N)}]")
