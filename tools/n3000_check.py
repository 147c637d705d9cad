"""Development aid: persistent n_fft = 3000 kernel against the fp64 oracle (full, padded, truncated sequences, odd tile counts), then timing
(SPECTRE_MIXEDP=0 in a second run = the one-tile-per-workgroup kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fft_amd.functional import spectral_mix, describe, time_kernel
from oracle.spectral_mix_oracle import spectral_mix_numpy, assert_close
dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 3000
ok = True
if len(sys.argv) < 2 or sys.argv[1] != "time":
    for (B, Nin, D, G) in [(1, 3000, 16, 1), (3, 3000, 64, 4), (5, 3000, 80, 5), (37, 3000, 112, 7), (2, 2900, 48, 3), (2, 700, 32, 2), (2, 3500, 32, 2), (24, 3000, 768, 4)]:
        V = torch.randn(B, Nin, D, device=dev)
        g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
        try:
            desc = describe(V, g, None, N)
            y = spectral_mix(V, g, None, N); torch.cuda.synchronize()
            assert y.shape == (B, min(Nin, N), D)
            if B * D <= 4096:
                ref = spectral_mix_numpy(V.cpu().numpy(), g.cpu().numpy(), None, N)
                err = assert_close(y.cpu().numpy(), ref, what="n3000")
            else:
                err = 0.0
                for (b, c) in [(0, 0), (B - 1, D - 1), (B // 2, 17), (7, D // 2 + 1), (B - 2, (16 * 13 + 5) % D)]:
                    c0 = c // 2 * 2
                    grp = c0 // (D // G)
                    ref = spectral_mix_numpy(V[b:b+1, :, c0:c0+2].cpu().numpy(), g[b:b+1, grp:grp + 1].cpu().numpy(), None, N)
                    err = max(err, assert_close(y[b:b+1, :, c0:c0+2].cpu().numpy(), ref, what="n3000 col"))
            print(f"OK   ({B},{Nin},{D}) G={G} err/rms={err:.2e} [{desc}]")
        except Exception as e:
            ok = False
            print(f"FAIL ({B},{Nin},{D}) G={G}: {type(e).__name__}: {str(e)[:300]}")
    print("PARITY", "OK" if ok else "FAILED")
B, D, G = 256, 768, 4
V = torch.randn(B, N, D, device=dev)
g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
out = torch.empty_like(V)
byt = B * N * D * 8 + B * G * (N // 2 + 1) * 8
ms = min(time_kernel(V, g, None, N, out=out, warmup=3, iters=10) for _ in range(3))
print(f"TIME (256,3000,768) f32: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  frac={byt/ms/1e6/8000:.3f} [{describe(V, g, None, N)[:40]}]")
