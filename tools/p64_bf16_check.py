"""Development aid: the pipelined 4096 kernel with bf16 rows in / fp32 rows out against the fp64 oracle (the oracle is given the
bf16-rounded input, so only the arithmetic is compared), full, padded and truncated sequences; then timing with and without it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fft_amd.functional import spectral_mix, describe, time_kernel
from oracle.spectral_mix_oracle import spectral_mix_numpy, assert_close
dev = torch.device("cuda:0")
torch.manual_seed(0)
ok = True
for (B, Nin, D, G) in [(1, 4096, 16, 1), (3, 4096, 64, 4), (5, 4096, 80, 5), (37, 4096, 112, 7), (3, 4000, 64, 2), (2, 1000, 48, 3), (2, 5000, 32, 2), (40, 4096, 768, 4)]:
    N = 4096
    V = torch.randn(B, Nin, D, device=dev).bfloat16()
    g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    try:
        desc = describe(V, g, None, N, out_dtype=torch.float32)
        y = spectral_mix(V, g, None, N, out_dtype=torch.float32); torch.cuda.synchronize()
        assert y.dtype == torch.float32 and y.shape == (B, min(Nin, N), D)
        Vf = V.float()
        if B * D <= 4096:
            ref = spectral_mix_numpy(Vf.cpu().numpy(), g.cpu().numpy(), None, N)
            err = assert_close(y.cpu().numpy(), ref, what="p64 bf16")
        else:
            err = 0.0
            for (b, c) in [(0, 0), (B - 1, D - 1), (B // 2, 17), (7, D // 2 + 1), (B - 2, (16 * 13 + 5) % D)]:
                c0 = c // 2 * 2
                grp = c0 // (D // G)
                ref = spectral_mix_numpy(Vf[b:b+1, :, c0:c0+2].cpu().numpy(), g[b:b+1, grp:grp + 1].cpu().numpy(), None, N)
                err = max(err, assert_close(y[b:b+1, :, c0:c0+2].cpu().numpy(), ref, what="p64 bf16 col"))
        print(f"OK   ({B},{Nin},{D}) G={G} err/rms={err:.2e} [{desc}]")
    except Exception as e:
        ok = False
        print(f"FAIL ({B},{Nin},{D}) G={G}: {type(e).__name__}: {str(e)[:300]}")
print("PARITY", "OK" if ok else "FAILED")
if len(sys.argv) > 1 and sys.argv[1] == "time":
    B, N, D, G = 256, 4096, 768, 4
    V = torch.randn(B, N, D, device=dev).bfloat16()
    g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    out = torch.empty(B, N, D, device=dev)
    byt = B * N * D * 6 + B * G * (N // 2 + 1) * 8
    for rep in range(3):
        ms = time_kernel(V, g, None, N, out=out, warmup=3, iters=10)
        print(f"TIME (256,4096,768) bf16->f32: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  frac={byt/ms/1e6/8000:.3f} [{describe(V, g, None, N, out_dtype=torch.float32)}]")
