"""n_fft = 4096 fp32: requests spread over the arithmetic, (SPLIT, PF) = (3, 3) (round 4, third session) against the phased order of the
previous build, through the LIBRARY: ab_libs/libspectre_r04_before_spread.so against the current one, each in its own process, interleaved."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel, spectral_mix
dev = "cuda:0"
B, N, D = 256, 4096, 768
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
res = []
for k in range(3):                      # three (V, out) allocation pairs: the time belongs to the pair (LABNOTES section 5 item 7)
    Vv = V.clone(); out = torch.empty(B, N, D, device=dev)
    ms = min(time_kernel(Vv, g, None, N, out=out, warmup=40, iters=20) for _ in range(2))
    ref = spectral_mix(Vv[:1], g[:1], None, N, algo="stockham")
    res.append("%%.4f (%%.0e)" %% (ms, float((out[:1] - ref).abs().max())))
    keep = (Vv, out) if k == 0 else keep
print("MS " + "  ".join(res))
''' % ROOT
old = os.path.join(ROOT, "ab_libs", "libspectre_r04_before_spread.so")
for r in range(3):
    for name, env in (("spread", {}), ("phased", {"SPECTRE_HIP_LIB": old})):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("%-8s" % name, [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-400:] if out.returncode else "", flush=True)
