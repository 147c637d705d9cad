"""Launch the headline spectral mix a few times (target of the rocprofv3 passes in tools/profile.sh)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import spectral_mix, describe

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256,4096,768")
ap.add_argument("--io", default="f32")
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--algo", default="auto")
a = ap.parse_args()
B, N, D = (int(x) for x in a.shape.split(","))
dt = torch.float32 if a.io == "f32" else torch.bfloat16
torch.manual_seed(0)
V = torch.randn(B, N, D, device="cuda").to(dt)
F = N // 2 + 1
gate = torch.randn(B, 4, F, dtype=torch.complex64, device="cuda") * 0.3
gate = gate * (torch.rand(B, 4, F, device="cuda") >= 0.18)
out = torch.empty_like(V)
print(describe(V, gate, None, N, algo=a.algo))
for _ in range(a.iters):
    spectral_mix(V, gate, None, N, out=out, algo=a.algo)
torch.cuda.synchronize()
