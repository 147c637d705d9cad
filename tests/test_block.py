"""The outermost caller of the path (SURVEY.md section 8(b) "Callers"): SpectreBlock, spectre.py:892-982 — residuals, LayerNorms, MLP and the
block's spectral memory (row a4: source parameter :951-959, zero-padded to all bins :973-977, per-head chunks :706-707).
CPU: surface + reference state_dicts + the padded-memory cache; GPU: the outputs and autograd results the REFERENCE produced (fixtures g12_*,
tests/golden/make_golden.py `case_block`), incl. the gradient of an un-frozen memory through the library's own rfft launch."""
import glob
import inspect
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden
from oracle.spectral_mix_oracle import assert_close

BLK = sorted(glob.glob(os.path.join(GOLDEN_DIR, "g12_block_*.npz")))
BLKG = [p for p in BLK if "dout" in np.load(p).files]


def _build(d):
    from fft_amd import SpectreBlock
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}
    E = sd["ln1.weight"].shape[0]
    blk = SpectreBlock(E, int(d["H"]), int(d["n_fft"]), pooling_type="mean", num_groups=int(d["G"]), wavelet_on_rate=0.0,
                       memory_size=int(d["memory_size"])).eval()
    missing, unexpected = blk.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return blk


def test_surface_and_reference_state_dicts():
    from fft_amd import SpectreBlock
    names = list(inspect.signature(SpectreBlock.__init__).parameters)
    assert names == ["self", "embed_dim", "num_heads", "n_fft", "mlp_ratio", "d_gate", "use_toeplitz", "dropout_p", "pooling_type",
                     "num_groups", "num_buckets", "wavelet_on_rate", "memory_size"]                     # spectre.py:909-923
    assert list(inspect.signature(SpectreBlock.forward).parameters) == ["self", "x"]
    assert len(BLK) >= 5 and len(BLKG) >= 3
    for p in BLK:
        d = load_golden(p)
        blk = _build(d)
        F = int(d["n_fft"]) // 2 + 1
        ms = int(d["memory_size"])
        if ms == 0:
            assert blk.memory_fft is None and blk._memory_spectrum() is None
            continue
        assert blk.memory_fft.dtype == torch.complex64 and not blk.memory_fft.requires_grad                 # frozen, spectre.py:961
        assert blk.memory_fft.shape[0] == (F if ms == 1 else min(ms, F))                                      # spectre.py:949
        full = blk._memory_spectrum()
        assert full.shape == (F, blk.ln1.weight.shape[0]) and full.is_contiguous()
        bins = blk.memory_fft.shape[0]
        assert torch.equal(full[:bins], blk.memory_fft.detach()) and not full[bins:].abs().sum().item() > 0   # zero rows behind the stored bins, :973-977
        assert blk._memory_spectrum() is full                                                                # built once ...
        with torch.no_grad():
            blk.memory_fft.mul_(2.0)
        again = blk._memory_spectrum()
        assert again is not full and torch.equal(again[:bins], blk.memory_fft.detach())                      # ... per parameter version


def test_block_passes_the_wavelet_rate_to_the_layer_inside():
    from fft_amd import SpectreBlock
    assert SpectreBlock(32, 2, 64).mix.wavelet_refinement.on_rate == 0.1                  # spectre.py:921, :936
    assert SpectreBlock(32, 2, 64, wavelet_on_rate=0.3).mix.wavelet_refinement.on_rate == 0.3


def test_cpu_tensors_raise():
    d = load_golden(BLK[0])
    with pytest.raises(RuntimeError):
        _build(d)(torch.from_numpy(d["x"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", BLK, ids=[os.path.basename(p)[:-4] for p in BLK])
def test_forward_matches_reference(path):
    d = load_golden(path)
    blk = _build(d).to("cuda:0")
    x = torch.from_numpy(d["x"]).to("cuda:0")
    with torch.no_grad():
        y = blk(x)                                      # one spectral-mix launch over all heads, memory un-chunked
    y_graph = blk(x)                                    # parameters require grad: the autograd nodes
    torch.cuda.synchronize()
    assert tuple(y.shape) == d["out"].shape
    assert_close(y.cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="block forward")
    assert_close(y_graph.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="block forward (autograd path)")


@pytest.mark.gpu
@pytest.mark.parametrize("path", BLKG, ids=[os.path.basename(p)[:-4] for p in BLKG])
def test_gradients_match_reference_autograd(path):
    d = load_golden(path)
    blk = _build(d).to("cuda:0")
    train_memory = "grad/memory_fft" in d
    if train_memory:
        blk.memory_fft.requires_grad_(True)             # as the fixture's reference block was run
    x = torch.from_numpy(d["x"]).to("cuda:0").requires_grad_(True)
    out = blk(x)
    (out * torch.from_numpy(d["dout"]).to("cuda:0")).sum().backward()
    torch.cuda.synchronize()
    assert_close(out.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="forward under autograd")
    assert_close(x.grad.cpu().numpy(), d["grad_x"], rtol=1e-4, atol_rms=5e-4, what="d/dx")
    checked = 0
    for name, prm in blk.named_parameters():
        key = "grad/" + name
        if key not in d:
            assert prm.grad is None or not prm.grad.abs().max().item() > 0 or name.startswith("mix.wavelet_refinement"), name
            continue
        assert prm.grad is not None, name
        g, e = prm.grad.cpu().numpy(), d[key]
        if np.iscomplexobj(e):                          # d/d(memory_fft): (bins, E) complex — compare both planes
            assert g.shape == e.shape
            g, e = np.stack((g.real, g.imag)), np.stack((e.real, e.imag))
        assert_close(g, e, rtol=1e-4, atol_rms=1e-3, what="d/d " + name)
        checked += 1
    assert checked >= 8 * blk.mix.num_heads + 8         # the layer's parameters + ln1, ln2, mlp (8 tensors)
    assert ("grad/memory_fft" in d) == train_memory and (not train_memory or blk.memory_fft.grad is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,N", [(64, 64), (64, 40), (45, 45), (60, 70)])
def test_memory_gradient_against_the_oracle_autograd(n_fft, N):
    """spectral_memory_grad alone: d/d(mem) of sum(out * dout) with out = irfft(gate * rfft(V) + mem)[:, :N] — the oracle's torch restatement
    differentiated by autograd on the CPU against the HIP path's (batch sum + rfft launch + bin weights)."""
    from fft_amd import spectral_mix
    from fft_amd.functional import spectral_memory_grad
    from oracle.spectral_mix_oracle import spectral_mix_torch
    g = torch.Generator().manual_seed(n_fft * 100 + N)
    B, D, G, F = 3, 16, 2, n_fft // 2 + 1
    V = torch.randn(B, N, D, generator=g)
    gate = torch.complex(torch.randn(B, G, F, generator=g), torch.randn(B, G, F, generator=g)).to(torch.complex64)
    mem = torch.complex(torch.randn(F, D, generator=g), torch.randn(F, D, generator=g)).to(torch.complex64).requires_grad_(True)
    out = spectral_mix_torch(V, gate, mem, n_fft)
    dout = torch.randn(out.shape, generator=g)
    (out * dout).sum().backward()
    got = spectral_memory_grad(dout.to("cuda:0"), n_fft).cpu().numpy()
    e = mem.grad.numpy()
    assert_close(np.stack((got.real, got.imag)), np.stack((e.real, e.imag)), rtol=1e-4, atol_rms=1e-4, what="d/d memory_fft")
    y = spectral_mix(V.to("cuda:0"), gate.to("cuda:0"), mem.detach().to("cuda:0"), n_fft)
    assert_close(y.cpu().numpy(), out.detach().numpy(), what="forward")
