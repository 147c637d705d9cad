"""Randomised parity stress of the persistent kernels (manual s_waitcnt counts, LDS-only barriers: a race would show up as a wrong
column only now and then): random (B, N_in, D, G, n_fft, dtype) against the fp64 oracle on sampled columns, many launches per shape."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fft_amd.functional import spectral_mix, describe
from oracle.spectral_mix_oracle import spectral_mix_numpy, assert_close
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(n_cases):
    n = rng.choice([4096, 4096, 4096, 3000, 2560, 2400, 3072, 3600, 3840])
    G = rng.choice([1, 2, 3, 4, 6])
    d_g = 16 * rng.randint(1, 12)
    D = G * d_g
    B = rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 120])
    if B * D > 40000: B = max(1, 40000 // D)
    Nin = rng.choice([n, n, n, rng.randint(1, n), n + rng.randint(1, 500)])
    dt = rng.choice([torch.float32, torch.float32, torch.bfloat16]) if n == 4096 else torch.float32
    out_dt = rng.choice([torch.float32, dt])
    mem = n == 4096 and dt == torch.float32 and rng.random() < 0.2
    torch.manual_seed(case)
    V = torch.randn(B, Nin, D, device=dev).to(dt)
    g = torch.randn(B, G, n // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
    g = g * (torch.rand(B, G, n // 2 + 1, device=dev) >= 0.15)
    m = torch.randn(n // 2 + 1, D, dtype=torch.complex64, device=dev) * 0.2 if mem else None
    desc = describe(V, g, m, n, out_dtype=out_dt)
    ys = [spectral_mix(V, g, m, n, out_dtype=out_dt) for _ in range(4)]
    torch.cuda.synchronize()
    ok = all(torch.equal(ys[0], y) for y in ys[1:])
    cols = [(rng.randrange(B), rng.randrange(D // 2) * 2) for _ in range(4)] + [(0, 0), (B - 1, D - 2)]
    tol = dict(rtol=1e-2, atol_rms=1e-2) if out_dt == torch.bfloat16 else {}
    try:
        for (b, c) in cols:
            ref = spectral_mix_numpy(V[b:b+1, :, c:c+2].float().cpu().numpy(), g[b:b+1, c // d_g:c // d_g + 1].cpu().numpy(),
                                     None if m is None else m[:, c:c+2].cpu().numpy(), n)
            assert_close(ys[0][b:b+1, :, c:c+2].float().cpu().numpy(), ref, what=f"case {case}", **tol)
    except AssertionError as e:
        ok = False; print("MISMATCH", str(e)[:200])
    bad += (not ok)
    print(("ok  " if ok else "BAD ") + f"case {case}: B={B} N_in={Nin} D={D} G={G} n_fft={n} {str(dt)[6:]}->{str(out_dt)[6:]} mem={mem} [{desc[:46]}]")
print("STRESS", "OK" if bad == 0 else f"FAILED ({bad})")
