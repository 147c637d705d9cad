"""Whole-line kernels (n_fft <= 1024): tile map of the non-persistent launch — XCD-contiguous (shipped) against blockIdx order and batch-major
(fft_amd/lib/libspectre_hip_wm1.so / _wm2.so = tools/build_variant.sh wmK regtile_wide.hip -DSPECTRE_WIDE_MAP=K), through the LIBRARY, one process
each, interleaved; three (V, out) pairs per process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch, hashlib
from fft_amd import time_kernel, spectral_mix, describe
dev = "cuda:0"
B, N, D = 256, int(sys.argv[1]), 768
dt = torch.bfloat16 if sys.argv[2] == "bf16" else torch.float32
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev).to(dt); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
res = []
for k in range(3):
    Vv = V.clone(); out = torch.empty(B, N, D, device=dev, dtype=dt)
    for _ in range(100): spectral_mix(Vv, g, None, N, out=out)
    ms = min(time_kernel(Vv, g, None, N, out=out, warmup=30, iters=60) for _ in range(3))
    res.append("%%.4f" %% ms)
print("MS " + "  ".join(res) + "  sha " + hashlib.sha1(out.view(torch.int16 if dt == torch.bfloat16 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:10] + "  " + describe(Vv, g, None, N, out=out)[:40])
''' % ROOT
libs = [("xcd-contig", {}), ("blockIdx", {"SPECTRE_HIP_LIB": os.path.join(ROOT, "fft_amd", "lib", "libspectre_hip_wm1.so")}),
        ("batch-major", {"SPECTRE_HIP_LIB": os.path.join(ROOT, "fft_amd", "lib", "libspectre_hip_wm2.so")})]
for n, io in ((1024, "f32"), (512, "f32"), (1024, "bf16")):
    for r in range(3):
        for name, env in libs:
            out = subprocess.run([sys.executable, "-c", CHILD, str(n), io], env=dict(os.environ, **env), capture_output=True, text=True)
            print("%5d %-5s %-11s" % (n, io, name), [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-300:] if out.returncode else "", flush=True)
