"""Quick on-GPU parity sweep (development aid; the real tests live in tests/)."""
import glob, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd.functional import spectral_mix, describe, time_kernel
from oracle.spectral_mix_oracle import spectral_mix_numpy, assert_close

dev = torch.device("cuda:0")
ok = True
for f in sorted(glob.glob("tests/golden/*.npz")):
    d = np.load(f)
    V = torch.from_numpy(d["V"]).to(dev); g = torch.from_numpy(d["gate"]).to(dev)
    mem = torch.from_numpy(d["mem"]).to(dev) if "mem" in d else None
    n = int(d["n_fft"])
    for algo in ("auto", "stockham"):
        try:
            desc = describe(V, g, mem, n, algo=algo)
            y = spectral_mix(V, g, mem, n, algo=algo); torch.cuda.synchronize()
            err = assert_close(y.cpu().numpy(), d["out"], what=f)
            print(f"OK   {os.path.basename(f):28s} {algo:9s} err/rms={err:.2e}  [{desc}]")
        except Exception as e:
            ok = False
            print(f"FAIL {os.path.basename(f):28s} {algo:9s} {type(e).__name__}: {str(e)[:300]}")
# bigger shapes vs numpy float64 oracle
torch.manual_seed(0)
for (B, N, D, G, n_fft, dt) in [(3, 4096, 64, 4, 4096, torch.float32), (2, 1024, 48, 3, 1024, torch.float32),
                                (2, 256, 32, 2, 256, torch.float32), (2, 3000, 64, 4, 3000, torch.float32),
                                (2, 4096, 32, 2, 4096, torch.bfloat16), (2, 1000, 32, 2, 1024, torch.float32),
                                (2, 5000, 32, 2, 4096, torch.float32)]:
    V = torch.randn(B, N, D, device=dev).to(dt)
    F = n_fft // 2 + 1
    g = (torch.randn(B, G, F, device=dev, dtype=torch.complex64) * 0.3)
    mem = torch.randn(F, D, device=dev, dtype=torch.complex64) * 0.2
    for m in (None, mem):
        for algo in ("auto", "stockham"):
            try:
                desc = describe(V, g, m, n_fft, algo=algo, out_dtype=torch.float32)
                y = spectral_mix(V, g, m, n_fft, algo=algo, out_dtype=torch.float32); torch.cuda.synchronize()
                ref = spectral_mix_numpy(V.float().cpu().numpy(), g.cpu().numpy(), None if m is None else m.cpu().numpy(), n_fft)
                err = assert_close(y.cpu().numpy(), ref, what="big")
                print(f"OK   ({B},{N},{D}) G={G} n_fft={n_fft} {str(dt)[6:]:8s} mem={m is not None} {algo:9s} err/rms={err:.2e} [{desc}]")
            except Exception as e:
                ok = False
                print(f"FAIL ({B},{N},{D}) G={G} n_fft={n_fft} {dt} mem={m is not None} {algo}: {type(e).__name__}: {str(e)[:300]}")
# headline timing
for dt in (torch.float32, torch.bfloat16):
    B, N, D, G = 256, 4096, 768, 4
    V = torch.randn(B, N, D, device=dev).to(dt)
    g = torch.randn(B, G, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    out = torch.empty_like(V)
    ms = time_kernel(V, g, None, N, out=out, warmup=3, iters=10)
    es = V.element_size()
    byt = B * N * D * es * 2 + B * G * (N // 2 + 1) * 8
    print(f"TIME {dt} (256,4096,768): {ms:.3f} ms  {B*N/ms/1e3:.1f} Mtok/s  {byt/ms/1e6:.0f} GB/s  ({byt/ms/1e6/8000*100:.1f}% of 8 TB/s) [{describe(V,g,None,N)}]")
    del V, g, out
for (B, N, D) in [(256, 1024, 768), (256, 3000, 768)]:
    V = torch.randn(B, N, D, device=dev)
    g = torch.randn(B, 4, N // 2 + 1, device=dev, dtype=torch.complex64) * 0.3
    out = torch.empty_like(V)
    ms = time_kernel(V, g, None, N, out=out, warmup=2, iters=5)
    byt = B * N * D * 8 + B * 4 * (N // 2 + 1) * 8
    print(f"TIME f32 ({B},{N},{D}): {ms:.3f} ms  {B*N/ms/1e3:.1f} Mtok/s  {byt/ms/1e6:.0f} GB/s [{describe(V,g,None,N)}]")
print("ALL OK" if ok else "SOME FAILED")
