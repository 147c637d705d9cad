import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev="cuda:0"
for (B,N,D,n) in [(64,8192,768,8192),(64,8000,768,8192),(32,16384,768,16384),(32,16000,768,16384)]:
    V=torch.randn(B,N,D,device=dev); g=torch.randn(B,4,n//2+1,dtype=torch.complex64,device=dev)*0.3
    out=torch.empty(B,min(N,n),D,device=dev)
    ms=min(time_kernel(V,g,None,n,out=out,warmup=2,iters=6) for _ in range(3))
    print(f"({B},{N},{D}) n_fft={n}: {ms:.3f} ms [{describe(V,g,None,n)[:50]}]")
