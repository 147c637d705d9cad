import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fft_amd import describe, spectral_mix
DEV="cuda:0"
g0 = torch.Generator().manual_seed(3)
B,N,D,G=12,4096,64,4
V = torch.randn(B,N,D,generator=g0).to(DEV)
gate = (torch.complex(torch.randn(B,G,N//2+1,generator=g0), torch.randn(B,G,N//2+1,generator=g0))*0.3).to(torch.complex64).to(DEV)
out = torch.empty_like(V)
for pre in range(int(sys.argv[1]) if len(sys.argv)>1 else 1):
    spectral_mix(V, gate, None, 4096, out=out)
torch.cuda.synchronize()
print("before capture:", describe(V, gate, None, 4096, out=out))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    spectral_mix(V, gate, None, 4096, out=out)
print("after capture:", describe(V, gate, None, 4096, out=out))
for seed in (7,8,9):
    V.copy_(torch.randn(V.shape, generator=torch.Generator().manual_seed(seed)).to(DEV))
    out.zero_()
    g.replay(); torch.cuda.synchronize()
    ref = spectral_mix(V, gate, None, 4096); torch.cuda.synchronize()
    bad = (out != ref)
    print("seed", seed, "mismatching elements", int(bad.sum()), "zeros in out", int((out==0).sum()))
    if bad.any():
        bt = bad.view(B, N, D//16, 16).any(dim=3)          # (B, N, tiles)
        per_tile = bt.sum(dim=1)                            # rows bad per (b, tile)
        print(" rows bad per (batch, column tile):", per_tile.flatten().tolist())
        b, t = [int(x) for x in (per_tile == per_tile.max()).nonzero()[0]]
        rows = bt[b, :, t].nonzero().flatten()
        print(" example tile", b, t, "bad rows (first 40):", rows[:40].tolist(), "row mod 64 set:", sorted(set((rows % 64).tolist()))[:70], "row//64 mod 8 set:", sorted(set(((rows//64)%8).tolist())))
