"""Gate gradient at n_fft = 4096, fast mode: the PERSISTENT form with prefetch registers (kernel_regtile_grad.h, PN / EARLY; round 5).
A workgroup walks through several (batch, group, s) work items, requests the first tile of the next item during the last tile of the
current one, and flushes / clears its LDS accumulator at every item boundary — the cases below make those boundaries ragged:
more items than workgroups with a remainder, items of one and of two tiles in the same launch, one-tile items only (every tile is a
first tile), bf16 rows.  Whole tensors against the float64 closed form; full size against the form without prefetch registers
(subprocess: the tuning switch is read once per process) and launch-to-launch determinism."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle.spectral_mix_oracle import assert_close, spectral_mix_backward_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,D,G,dtype", [(40, 80, 2, torch.float32),      # d_g = 40: T = 5 tiles, S = 4 -> 320 items of 2 / 1 / 1 / 1 tiles on 256 workgroups
                                         (70, 64, 4, torch.bfloat16),     # d_g = 16: T = 2, S = 2 -> 560 one-tile items: 2.2 rounds
                                         (3, 768, 4, torch.float32),      # fewer items than workgroups: one item each, 6 tiles
                                         (65, 96, 1, torch.float32)])     # d_g = 96: T = 12, S = 4 -> 260 items, 4 workgroups take a second one
def test_persistent_gate_gradient_whole_tensor(B, D, G, dtype):
    from fft_amd import spectral_mix_backward
    N = 4096
    g = torch.Generator().manual_seed(B * 1000 + D)
    V = torch.randn(B, N, D, generator=g).to(dtype)
    dY = torch.randn(B, N, D, generator=g).to(dtype)
    gate = (torch.complex(torch.randn(B, G, N // 2 + 1, generator=g), torch.randn(B, G, N // 2 + 1, generator=g)) * 0.3).to(torch.complex64)
    _, dG = spectral_mix_backward(V.to(DEV), gate.to(DEV), dY.to(DEV), N, need_dv=False)
    _, dG2 = spectral_mix_backward(V.to(DEV), gate.to(DEV), dY.to(DEV), N, need_dv=False)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(dG), torch.view_as_real(dG2)), "two launches differ"
    _, ref = spectral_mix_backward_numpy(V.float().numpy(), gate.numpy(), dY.float().numpy(), N)
    assert_close(torch.view_as_real(dG).cpu().numpy(), np.stack([ref.real, ref.imag], -1), what=f"dgate ({B},{N},{D}) G={G} {dtype}")


CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from fft_amd import spectral_mix_backward
B, N, D, G = 256, 4096, 768, 4
torch.manual_seed(3)
V = torch.randn(B, N, D, device="cuda:0"); dY = torch.randn(B, N, D, device="cuda:0")
gate = torch.randn(B, G, N // 2 + 1, dtype=torch.complex64, device="cuda:0") * 0.3
outs = {}
for pf in ("0", "1"):
    os.environ["SPECTRE_DGATE_PREFETCH"] = pf
    outs[pf] = spectral_mix_backward(V, gate, dY, N, need_dv=False)[1].clone()
torch.cuda.synchronize()
a, b = outs["0"], outs["1"]
print("RESULT", float((a - b).abs().max() / a.abs().max()), float(a.abs().max()), bool(torch.isfinite(torch.view_as_real(b)).all()))
''' % ROOT


def test_full_size_against_the_form_without_prefetch_registers():
    """(256, 4096, 768), G = 4: 4096 work items on 256 persistent workgroups, 16 each.  Same sums in the same order — the two forms may
    differ by the compiler's choice of fused multiply-adds only."""
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SPECTRE_TUNING="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")][0].split()
    rel, amax, finite = float(line[1]), float(line[2]), line[3] == "True"
    assert finite and amax > 1.0
    assert rel < 2e-6, f"persistent form differs from the one-item-per-workgroup form by {rel:.2e} of the largest entry"
