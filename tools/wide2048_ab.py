"""n_fft = 2048: one whole-line (32-channel) workgroup per CU (SPECTRE_WIDE_MAX=2048) against two 16-channel workgroups per CU."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel, describe, spectral_mix
dev = "cuda:0"
B, N, D = 256, 2048, 768
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3; out = torch.empty_like(V)
ms = min(time_kernel(V, g, None, N, out=out, warmup=30, iters=20) for _ in range(3))
ref = spectral_mix(V[:2], g[:2], None, N, algo="stockham")
err = float((out[:2] - ref).abs().max())
byt = 2 * B * N * D * 4 + B * 4 * (N // 2 + 1) * 8
print("2048 %%.4f ms %%.3f of 8 TB/s maxdiff vs stockham %%.2e [%%s]" %% (ms, byt / ms / 1e6 / 8000, err, describe(V, g, None, N)[:28]))
''' % ROOT
for r in range(3):
    for env in ({"SPECTRE_TUNING": "1", "SPECTRE_WIDE_MAX": "2048"}, {}):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("wide" if env else "16ch", "|", " | ".join(l for l in out.stdout.splitlines() if " ms " in l))
        if out.returncode: print(out.stderr[-600:])
