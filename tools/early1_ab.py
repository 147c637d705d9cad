"""n_fft = 4096: stage 1 of the deferred groups behind the store burst (SPECTRE_P64_EARLY1, kernel_regtile64p.h) — the shipped mask against
another one (SPECTRE_EARLY1_OLD_LIB, default fft_amd/lib/libspectre_hip_e0.so = tools/build_variant.sh e0 regtile_n4096p.hip
-DSPECTRE_P64_EARLY1=0), through the LIBRARY, one process each, interleaved; per process three (V, out) pairs, tile order pinned (static /
tickets) or measured (auto).    python tools/early1_ab.py [bf16out bf16 f32]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from fft_amd import time_kernel, spectral_mix, describe
dev = "cuda:0"
B, N, D = 256, 4096, 768
odt = torch.bfloat16 if sys.argv[1] == "bf16out" else torch.float32
idt = torch.float32 if sys.argv[1] == "f32" else torch.bfloat16
torch.manual_seed(0)
V = torch.randn(B, N, D, device=dev).to(idt); g = torch.randn(B, 4, N // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
res = []
for k in range(3):
    Vv = V.clone(); out = torch.empty(B, N, D, device=dev, dtype=odt)
    for _ in range(50): spectral_mix(Vv, g, None, N, out=out)
    ms = min(time_kernel(Vv, g, None, N, out=out, warmup=20, iters=30) for _ in range(2))
    res.append("%%.4f" %% ms)
    keep = (Vv, out) if k == 0 else keep
import hashlib
print("MS " + "  ".join(res) + "  sha " + hashlib.sha1(out.view(torch.int16 if odt == torch.bfloat16 else torch.int32).cpu().numpy().tobytes()).hexdigest()[:10] + "  " + describe(Vv, g, None, N, out=out)[-60:])
''' % ROOT
old = os.environ.get("SPECTRE_EARLY1_OLD_LIB") or os.path.join(ROOT, "fft_amd", "lib", "libspectre_hip_e0.so")
for io in (sys.argv[1:] or ["bf16out", "bf16"]):
    for order in ("static", "tickets", None):
        for r in range(2):
            for name, env in (("shipped", {}), ("other", {"SPECTRE_HIP_LIB": old})):
                e = dict(os.environ, **env)
                if order: e.update(SPECTRE_TUNING="1", SPECTRE_TILE_ORDER=order)
                out = subprocess.run([sys.executable, "-c", CHILD, io], env=e, capture_output=True, text=True)
                print("%-8s %-8s %-7s" % (io, order or "auto", name), [l for l in out.stdout.splitlines() if l.startswith("MS")], out.stderr[-300:] if out.returncode else "", flush=True)
