"""Development aid: parity of the pipelined 4096 kernel against the fp64 oracle on shapes that exercise the persistent loop
(odd tile counts, several tiles per workgroup, one tile, conj gate through the backward), then headline timing A/B."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fft_amd.functional import spectral_mix, describe, time_kernel
from oracle.spectral_mix_oracle import spectral_mix_numpy, assert_close
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
for (B, D, G) in [(1, 16, 1), (3, 64, 4), (2, 48, 3), (5, 80, 5), (37, 112, 7), (40, 768, 4)]:
    N = 4096
    V = torch.randn(B, N, D, device=dev)
    g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    try:
        desc = describe(V, g, None, N)
        y = spectral_mix(V, g, None, N); torch.cuda.synchronize()
        if B * D <= 4096:
            ref = spectral_mix_numpy(V.cpu().numpy(), g.cpu().numpy(), None, N)
            err = assert_close(y.cpu().numpy(), ref, what="p64")
        else:   # column spot check
            idx = [(0, 0), (B - 1, D - 1), (B // 2, 17), (7, D // 2 + 1), (B - 2, (16 * 13 + 5) % D)]
            err = 0.0
            for (b, c) in idx:
                c0 = c // 2 * 2
                grp = c0 // (D // G)
                ref = spectral_mix_numpy(V[b:b+1, :, c0:c0+2].cpu().numpy(), g[b:b+1, grp:grp + 1].cpu().numpy(), None, N)
                err = max(err, assert_close(y[b:b+1, :, c0:c0+2].cpu().numpy(), ref, what="p64 col"))
        print(f"OK   ({B},{N},{D}) G={G} err/rms={err:.2e} [{desc}]")
    except Exception as e:
        ok = False
        print(f"FAIL ({B},{N},{D}) G={G}: {type(e).__name__}: {str(e)[:300]}")
print("PARITY", "OK" if ok else "FAILED")
if len(sys.argv) > 1 and sys.argv[1] == "time":
    B, N, D, G = 256, 4096, 768, 4
    V = torch.randn(B, N, D, device=dev)
    g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    out = torch.empty_like(V)
    byt = B * N * D * 8 + B * G * (N // 2 + 1) * 8
    for rep in range(3):
        ms = time_kernel(V, g, None, N, out=out, warmup=3, iters=10)
        print(f"TIME (256,4096,768) f32: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  frac={byt/ms/1e6/8000:.3f} [{describe(V,g,None,N)}]")
    for (tin, tout) in ((torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)):
        Vb = V.to(tin); ob = torch.empty(B, N, D, dtype=tout, device=dev)
        bytb = B * N * D * (Vb.element_size() + ob.element_size()) + B * G * (N // 2 + 1) * 8
        for rep in range(2):
            ms = time_kernel(Vb, g, None, N, out=ob, warmup=3, iters=10)
            print(f"TIME (256,4096,768) {str(tin)[6:]}->{str(tout)[6:]}: {ms:.3f} ms  {bytb/ms/1e6:.0f} GB/s  frac={bytb/ms/1e6/8000:.3f} [{describe(Vb,g,None,N,out_dtype=tout)}]")
