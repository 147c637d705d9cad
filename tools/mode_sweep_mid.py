import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fft_amd import describe, time_kernel
dev="cuda:0"
for (B,N,D,n,mem) in [(256,2048,768,2048,False),(256,2000,768,2048,False),(256,2048,768,2048,True),(256,512,768,512,False),(256,500,768,512,False),(256,1536,768,1536,False),(256,1500,768,1536,False),(256,2000,768,2000,False),(256,1990,768,2000,False)]:
    V=torch.randn(B,N,D,device=dev); g=torch.randn(B,4,n//2+1,dtype=torch.complex64,device=dev)*0.3
    m=torch.randn(n//2+1,D,dtype=torch.complex64,device=dev) if mem else None
    out=torch.empty(B,min(N,n),D,device=dev)
    ms=min(time_kernel(V,g,m,n,out=out,warmup=2,iters=6) for _ in range(3))
    print(f"({B},{N},{D}) n_fft={n} mem={mem}: {ms:.3f} ms [{describe(V,g,m,n)[:52]}]")
