import torch, sys
sys.path.insert(0, "/root/repo")
from fft_amd import spectral_mix_backward
dev = "cuda:0"
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, N, G = 256, 4096, 4
for D, expand in [(768, False), (768, True), (64, True), (64, False), (32, True)]:
    Gd = G if D >= 64 else 1
    F = N // 2 + 1
    if expand:
        V = torch.randn(1, N, D, device=dev).expand(B, N, D)
        dY = torch.randn(1, N, D, device=dev).expand(B, N, D)
    else:
        V = torch.randn(B, N, D, device=dev); dY = torch.randn(B, N, D, device=dev)
    gate = torch.randn(B, Gd, F, dtype=torch.complex64, device=dev)
    try:
        ms = t(lambda: spectral_mix_backward(V, gate, dY, N, need_dv=False))
        tiles = B * D // 8
        print(f"D={D} expand={expand}: {ms:.3f} ms, {tiles} tiles, {ms*1e3*256/tiles:.2f} us per tile per CU")
    except Exception as ex:
        print(D, expand, "ERR", str(ex)[:200])
