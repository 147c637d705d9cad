"""n_fft = 4096: store burst behind a workgroup barrier (round 4) against the round-3 store order, through the LIBRARY, each form in its own
process, interleaved, for every variant of the pipelined kernel.  SPECTRE_P64_BURST=0 (under SPECTRE_TUNING=1) = round-3 order.  (The run that decided what ships also had
"burst 2" = a second barrier in front of the deferred stores and "burst+meet" = the gang meeting of tools/p64v.h: profiles/r04_burst_ab.log.)"""
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
mem = (torch.randn(N // 2 + 1, D, dtype=torch.complex64, device=dev) * 0.2)
res = []
for name, tin, tout, m in (("f32", torch.float32, torch.float32, None), ("bf16->f32", torch.bfloat16, torch.float32, None), ("bf16->bf16", torch.bfloat16, torch.bfloat16, None), ("f32+mem", torch.float32, torch.float32, mem)):
    Vv = V.to(tin); out = torch.empty(B, N, D, dtype=tout, device=dev)
    ms = min(time_kernel(Vv, g, m, N, out=out, warmup=40, iters=20) for _ in range(2))
    ref = spectral_mix(Vv[:1], g[:1], m, N, algo="stockham", out_dtype=torch.float32)
    res.append("%%s %%.4f (%%.0e)" %% (name, ms, float((out[:1].float() - ref).abs().max())))
print("MS " + "  ".join(res))
''' % ROOT
for r in range(3):
    for name, env in (("burst", {}), ("round 3", {"SPECTRE_TUNING": "1", "SPECTRE_P64_BURST": "0"})):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("%-10s" % name, [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-400:] if out.returncode else "")
