"""The wavelet refinement launch (kernel_wavelet.h behind spectre_wavelet_refine / spectre_wavelet_gate_grad) against the oracle on seeded
inputs, against the reference's own outputs and autograd (fixtures g13_*), alone and inside the multi-head layer and the block."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_golden
from oracle.spectral_mix_oracle import assert_close, bf16_round
from oracle.wavelet_oracle import wavelet_gate_grad_numpy, wavelet_refinement_numpy

pytestmark = pytest.mark.gpu
WV = sorted(glob.glob(os.path.join(GOLDEN_DIR, "g13_wavelet_*.npz")))
LAYERS = sorted(glob.glob(os.path.join(GOLDEN_DIR, "g13_layer_*.npz")))
DEV = "cuda:0"


def _problem(B, N, D, seed, p_on=0.5):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, N, D, generator=g)
    gate = torch.rand(B, D, generator=g)
    mask = torch.rand(B, generator=g) < p_on
    if B > 1:
        mask[0], mask[-1] = True, False
    return v, gate, mask


@pytest.mark.parametrize("B,N,D", [(3, 1, 4), (3, 2, 8), (2, 4, 3), (4, 8, 16), (5, 64, 48), (3, 256, 40), (2, 1024, 17), (3, 2048, 32),
                                   (3, 4096, 24), (2, 8192, 12), (2, 16384, 6), (2, 32768, 3),
                                   # whole aligned tiles: level 0 in registers (P = 1, 2, 4, 8, 16 pairs per thread; 16 / 8 / 4 channels per tile)
                                   (3, 256, 16), (2, 512, 32), (2, 1024, 48), (3, 2048, 64), (3, 4096, 32), (2, 8192, 16), (2, 16384, 8)])
def test_kernel_against_oracle(B, N, D):
    from fft_amd import wavelet_refine
    v, gate, mask = _problem(B, N, D, seed=N + D)
    e = wavelet_refinement_numpy(v.numpy(), gate.numpy(), mask.numpy())
    vd = v.to(DEV)
    out, ref = wavelet_refine(vd, gate.to(DEV), mask.to(DEV), want_ref=True)
    torch.cuda.synchronize()
    assert out.data_ptr() != vd.data_ptr() and torch.equal(vd.cpu(), v)                   # out of place: the input is untouched
    assert_close(out.cpu().numpy(), e, rtol=1e-5, atol_rms=1e-5, what="refined")
    assert torch.equal(out.cpu()[~mask], v[~mask])                                        # switched-off elements: bit copies
    r = ref.cpu().numpy()[mask.numpy()]
    e_r = wavelet_refinement_numpy(np.zeros_like(v.numpy()), np.ones_like(gate.numpy()), mask.numpy()) * 0      # shape only
    e_r = wavelet_refinement_numpy(v.numpy(), np.ones_like(gate.numpy()), mask.numpy())[mask.numpy()] - v.numpy()[mask.numpy()]
    assert_close(r, e_r, rtol=1e-5, atol_rms=1e-5, what="round trip of the switched-on elements")
    inpl, none = wavelet_refine(vd, gate.to(DEV), mask.to(DEV), inplace=True)
    assert none is None and inpl.data_ptr() == vd.data_ptr() and torch.equal(inpl, out)    # in place: same bits


def test_strided_views_and_bf16():
    from fft_amd import wavelet_refine
    v, gate, mask = _problem(4, 512, 80, seed=5)
    big = torch.zeros(4, 512, 128)
    big[:, :, 16:96] = v
    view = big.to(DEV)[:, :, 16:96]                                                        # channel slice of a wider buffer: strided rows
    out, _ = wavelet_refine(view, gate.to(DEV), mask.to(DEV), inplace=True)
    torch.cuda.synchronize()
    e = wavelet_refinement_numpy(v.numpy(), gate.numpy(), mask.numpy())
    assert_close(out.cpu().numpy(), e, rtol=1e-5, atol_rms=1e-5, what="strided view, in place")
    big2 = torch.zeros(4, 512, 128)
    big2[:, :, 32:112] = v
    view2 = big2.to(DEV)[:, :, 32:112]                                                     # 80 = 5 x 16 channels at an aligned offset: the register form, strided
    out2, ref2 = wavelet_refine(view2, gate.to(DEV), mask.to(DEV), want_ref=True)
    assert_close(out2.cpu().numpy(), e, rtol=1e-5, atol_rms=1e-5, what="strided view, register form")
    assert torch.equal(wavelet_refine(view2, gate.to(DEV), mask.to(DEV), inplace=True)[0], out2)
    vb = v.to(torch.bfloat16)
    ob, rb = wavelet_refine(vb.to(DEV), gate.to(DEV), mask.to(DEV), want_ref=True)
    torch.cuda.synchronize()
    eb = wavelet_refinement_numpy(vb.float().numpy(), gate.numpy(), mask.numpy())
    got = ob.float().cpu().numpy()
    assert ob.dtype == torch.bfloat16 and np.abs(got - bf16_round(eb.astype(np.float32))).max() <= 2.0 ** -7 * np.abs(eb).max()
    assert torch.equal(ob.cpu()[~mask], vb[~mask])


def test_gate_gradient_kernel_against_oracle():
    from fft_amd import wavelet_refine
    from fft_amd.functional import wavelet_gate_grad
    v, gate, mask = _problem(5, 1024, 100, seed=9)
    dout = torch.randn(5, 1024, 100, generator=torch.Generator().manual_seed(10))
    _, ref = wavelet_refine(v.to(DEV), gate.to(DEV), mask.to(DEV), want_ref=True)
    got = wavelet_gate_grad(dout.to(DEV), ref, mask.to(DEV)).cpu().numpy()
    assert_close(got, wavelet_gate_grad_numpy(v.numpy(), dout.numpy(), mask.numpy()), rtol=1e-4, atol_rms=1e-5, what="d/d gate")
    assert not np.abs(got[~mask.numpy()]).max() > 0


def test_errors():
    from fft_amd import wavelet_refine
    v, gate, mask = (t.to(DEV) for t in _problem(2, 64, 8, seed=1))
    with pytest.raises(NotImplementedError, match="power-of-two"):
        wavelet_refine(torch.zeros(2, 48, 8, device=DEV), gate, mask)
    with pytest.raises(NotImplementedError, match="too long"):
        wavelet_refine(torch.zeros(1, 65536, 8, device=DEV), gate[:1], mask[:1])
    with pytest.raises(ValueError):
        wavelet_refine(v, gate[:, :4], mask)
    with pytest.raises(ValueError):
        wavelet_refine(v, gate, mask.float())
    with pytest.raises(RuntimeError):
        wavelet_refine(v.cpu(), gate.cpu(), mask.cpu())


def _module(d):
    from fft_amd import WaveletRefinement
    wr = WaveletRefinement(d["v"].shape[2], on_rate=float(d["on_rate"]))
    wr.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}, strict=True)
    wr.forced_mask = torch.from_numpy(d["mask"])                     # the coin flips of the reference's run
    return wr.to(DEV)


@pytest.mark.parametrize("path", WV, ids=[os.path.basename(p)[:-4] for p in WV])
def test_module_matches_reference_forward_and_autograd(path):
    d = load_golden(path)
    wr = _module(d)
    v = torch.from_numpy(d["v"]).to(DEV)
    q = torch.from_numpy(d["q_pool"]).to(DEV)
    with torch.no_grad():
        y = wr(v, q)
    assert y.data_ptr() != v.data_ptr()
    assert_close(y.cpu().numpy(), d["out"], rtol=1e-5, atol_rms=1e-5, what="forward")
    v.requires_grad_(True); q.requires_grad_(True)
    out = wr(v, q)
    (out * torch.from_numpy(d["dout"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert_close(out.detach().cpu().numpy(), d["out"], rtol=1e-5, atol_rms=1e-5, what="forward under autograd")
    assert np.array_equal(v.grad.cpu().numpy(), d["grad_v"])
    assert_close(q.grad.cpu().numpy(), d["grad_q_pool"], rtol=1e-4, atol_rms=1e-4, what="d/dq_pool")
    for name, prm in wr.named_parameters():
        assert_close(prm.grad.cpu().numpy(), d["grad/" + name], rtol=1e-4, atol_rms=1e-4, what="d/d " + name)


def test_the_coin_flip_is_the_references_draw():
    """Without a forced mask the module draws `torch.rand(B, 1, 1, device=v.device) < on_rate` (spectre.py:841) from the device's generator:
    re-seeding and drawing the same shape predicts which elements it refined."""
    from fft_amd import WaveletRefinement
    wr = WaveletRefinement(16, on_rate=0.4).to(DEV)
    v = torch.randn(32, 64, 16, device=DEV)
    q = torch.randn(32, 16, device=DEV)
    torch.manual_seed(1234)
    with torch.no_grad():
        y = wr(v, q)
    torch.manual_seed(1234)
    mask = (torch.rand(32, 1, 1, device=DEV) < 0.4).view(32)
    changed = (y != v).flatten(1).any(dim=1)
    assert 0 < int(mask.sum()) < 32 and torch.equal(changed, mask)


def _layer(d):
    from fft_amd import SpectreBlock, SpectreMultiHead
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}
    rate = float(d["on_rate"])
    if "ln1.weight" in sd:
        m = SpectreBlock(sd["ln1.weight"].shape[0], int(d["H"]), int(d["n_fft"]), pooling_type="mean", num_groups=int(d["G"]),
                         wavelet_on_rate=rate, memory_size=int(d["memory_size"])).eval()
        wr = m.mix.wavelet_refinement
    else:
        m = SpectreMultiHead(sd["out_proj.weight"].shape[0], int(d["H"]), int(d["n_fft"]), pooling_type="mean", num_groups=int(d["G"]),
                             wavelet_on_rate=rate).eval()
        wr = m.wavelet_refinement
    m.load_state_dict(sd, strict=True)
    wr.forced_mask = torch.from_numpy(d["mask"])
    return m.to(DEV), (m.mix if "ln1.weight" in sd else m)


@pytest.mark.parametrize("path", LAYERS, ids=[os.path.basename(p)[:-4] for p in LAYERS])
@pytest.mark.parametrize("fused", [True, False], ids=["one_mix_node", "per_head_loop"])
def test_layers_with_the_refinement_match_the_reference(path, fused):
    d = load_golden(path)
    m, mh = _layer(d)
    mh.fused_autograd = fused
    x = torch.from_numpy(d["x"]).to(DEV)
    with torch.no_grad():
        y = m(x)
    assert_close(y.cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="forward (inference path)")
    x.requires_grad_(True)
    out = m(x)
    (out * torch.from_numpy(d["dout"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert_close(out.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="forward under autograd")
    assert_close(x.grad.cpu().numpy(), d["grad_x"], rtol=1e-4, atol_rms=5e-4, what="d/dx")
    checked = 0
    for name, prm in m.named_parameters():
        key = "grad/" + name
        if key not in d:
            continue
        assert prm.grad is not None, name
        assert_close(prm.grad.cpu().numpy(), d[key], rtol=1e-4, atol_rms=1e-3, what="d/d " + name)
        checked += "wavelet_refinement" in name
    assert checked == 4                                             # the refinement's gate MLP learns (two Linear layers), as in the reference


def test_block_under_autocast_and_in_a_graph():
    """bf16 autocast (per-head path, bf16 rows through the mix and the refinement, fp32 gate) stays close to the fp32 block; and the default
    block's inference forward — coin flip included — replays from a hipGraph (the mask is drawn and read on the device: nothing looks at it on the host)."""
    from fft_amd import SpectreBlock
    torch.manual_seed(3)
    blk = SpectreBlock(64, 2, 256, pooling_type="mean", num_groups=2, wavelet_on_rate=0.5, memory_size=9).to(DEV).eval()
    x = torch.randn(4, 256, 64, device=DEV)
    mask = torch.tensor([True, False, True, True])
    blk.mix.wavelet_refinement.forced_mask = mask
    with torch.no_grad():
        want = blk(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = blk(x)
    assert torch.isfinite(got).all()
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 3e-2, err
    xg = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        blk(xg).float().square().mean().backward()
    assert torch.isfinite(xg.grad).all() and all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in blk.named_parameters() if p.requires_grad)
    # graph capture with the DRAWN mask
    blk.mix.wavelet_refinement.forced_mask = None
    with torch.no_grad():
        blk(x); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = blk(x)
        outs = []
        for _ in range(6):
            g.replay(); torch.cuda.synchronize()
            outs.append(y.clone())
    assert all(torch.isfinite(o).all() for o in outs)
    assert any(not torch.equal(outs[0], o) for o in outs[1:])            # a fresh coin flip per replay (the generator advances under the graph)
