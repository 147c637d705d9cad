"""Row N3: the multi-head wrapper.  CPU: surface + reference state_dict; GPU: reference outputs (fixtures g11_*)."""
import glob
import inspect
import os

import pytest
import torch

from conftest import GOLDEN_DIR, load_golden
from oracle.spectral_mix_oracle import assert_close

MH = sorted(glob.glob(os.path.join(GOLDEN_DIR, "g11_multihead_*.npz")))


def _build(d):
    from fft_amd import SpectreMultiHead
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd/")}
    E = sd["out_proj.weight"].shape[0]
    mh = SpectreMultiHead(E, int(d["H"]), int(d["n_fft"]), pooling_type="mean", num_groups=int(d["G"]), wavelet_on_rate=0.0).eval()
    missing, unexpected = mh.load_state_dict(sd, strict=True)        # incl. the (unused) wavelet_refinement parameters
    assert not missing and not unexpected
    return mh


def test_surface_and_reference_state_dict():
    from fft_amd import SpectreMultiHead
    names = list(inspect.signature(SpectreMultiHead.__init__).parameters)
    assert names == ["self", "embed_dim", "num_heads", "n_fft", "d_gate", "use_toeplitz", "dropout_p", "pooling_type",
                     "num_groups", "num_buckets", "wavelet_on_rate"]                    # spectre.py:664-676
    assert list(inspect.signature(SpectreMultiHead.forward).parameters) == ["self", "x", "pos_phase", "memory_fft"]
    assert SpectreMultiHead(32, 2, 64).wavelet_refinement.on_rate == 0.1               # the reference's default (spectre.py:675); tests/test_wavelet_*.py
    assert SpectreMultiHead(32, 2, 64, wavelet_on_rate=0.0).wavelet_refinement.on_rate == 0.0
    assert len(MH) >= 2
    for p in MH:
        mh = _build(load_golden(p))
        assert mh.num_heads == len(mh.heads) and mh.head_dim * mh.num_heads == mh.out_proj.in_features


@pytest.mark.gpu
@pytest.mark.parametrize("path", MH, ids=[os.path.basename(p)[:-4] for p in MH])
def test_forward_matches_reference(path):
    d = load_golden(path)
    mh = _build(d).to("cuda:0")
    x = torch.from_numpy(d["x"]).to("cuda:0")
    pp = torch.from_numpy(d["pos_phase"]).to("cuda:0") if "pos_phase" in d else None
    mem = torch.from_numpy(d["mem"]).to("cuda:0") if "mem" in d else None
    with torch.no_grad():
        y = mh(x, pos_phase=pp, memory_fft=mem)                     # single-launch path: one spectral mix over all heads
    y_graph = mh(x, pos_phase=pp, memory_fft=mem)                   # autograd path: one value node + ONE mix node for all heads
    mh.fused_autograd = False
    y_loop = mh(x, pos_phase=pp, memory_fft=mem)                    # the reference's structure: per-head modules + cat
    torch.cuda.synchronize()
    assert tuple(y.shape) == d["out"].shape
    assert_close(y.cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="multi-head forward")
    assert_close(y_graph.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="multi-head forward (fused autograd path)")
    assert_close(y_loop.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="multi-head forward (per-head autograd path)")
    y_graph.sum().backward()
    assert all(p.grad is not None for n, p in mh.named_parameters() if not n.startswith("wavelet_refinement"))


MHG = [p for p in MH if "grad" in os.path.basename(p)]


@pytest.mark.gpu
@pytest.mark.parametrize("path", MHG, ids=[os.path.basename(p)[:-4] for p in MHG])
@pytest.mark.parametrize("fused", [True, False], ids=["one_mix_node", "per_head_loop"])
def test_gradients_match_reference_autograd(path, fused):
    """Training half of row N3: d/dx and d/d(parameter) of the multi-head layer for the fixture's upstream gradient equal what the
    reference's autograd produced (spectre.py:701-726 differentiated by torch) — through ONE spectral-mix node over all heads
    (fused=True, the default) and through the per-head modules."""
    d = load_golden(path)
    mh = _build(d).to("cuda:0")
    mh.fused_autograd = fused
    x = torch.from_numpy(d["x"]).to("cuda:0").requires_grad_(True)
    pp = torch.from_numpy(d["pos_phase"]).to("cuda:0") if "pos_phase" in d else None
    mem = torch.from_numpy(d["mem"]).to("cuda:0") if "mem" in d else None
    out = mh(x, pos_phase=pp, memory_fft=mem)
    (out * torch.from_numpy(d["dout"]).to("cuda:0")).sum().backward()
    torch.cuda.synchronize()
    assert_close(out.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="forward under autograd")
    assert_close(x.grad.cpu().numpy(), d["grad_x"], rtol=1e-4, atol_rms=5e-4, what="d/dx")
    checked = 0
    for name, prm in mh.named_parameters():
        key = "grad/" + name
        if key not in d:
            assert prm.grad is None or not prm.grad.abs().max().item() > 0 or name.startswith("wavelet_refinement"), name
            continue
        assert prm.grad is not None, name
        assert_close(prm.grad.cpu().numpy(), d[key], rtol=1e-4, atol_rms=1e-3, what="d/d " + name)
        checked += 1
    assert checked >= 8 * mh.num_heads                              # W_q, W_v, gate MLP (4), LayerNorm (2), modReLU bias per head + out_proj


@pytest.mark.gpu
def test_training_is_one_mix_node_over_all_heads(monkeypatch):
    """Under autograd the H heads are ONE spectral-mix launch forward (and one dV + one dgate launch backward), no per-head loop."""
    import fft_amd.spectre as sp
    d = load_golden(MHG[0])
    mh = _build(d).to("cuda:0")
    fwd, bwd = [], []
    real_f, real_b = sp.spectral_mix, sp.spectral_mix_backward
    monkeypatch.setattr(sp, "spectral_mix", lambda V, gate, *a, **k: (fwd.append((tuple(V.shape), tuple(gate.shape))), real_f(V, gate, *a, **k))[1])
    monkeypatch.setattr(sp, "spectral_mix_backward", lambda V, gate, *a, **k: (bwd.append(tuple(gate.shape)), real_b(V, gate, *a, **k))[1])
    x = torch.from_numpy(d["x"]).to("cuda:0").requires_grad_(True)
    mh(x).square().sum().backward()
    torch.cuda.synchronize()
    assert len(fwd) == 1 and len(bwd) == 1
    assert fwd[0][0][2] == mh.num_heads * mh.head_dim and fwd[0][1][1] == mh.num_heads * mh.heads[0].G == bwd[0][1]


@pytest.mark.gpu
def test_inference_is_one_mix_launch_over_all_heads(monkeypatch):
    """Row N3 as SURVEY.md wrote it: without autograd the H heads run as ONE spectral-mix call on a (B, H*G, F) gate."""
    import fft_amd.spectre as sp
    d = load_golden(MH[0])
    mh = _build(d).to("cuda:0")
    calls = []
    real = sp.spectral_mix

    def counting(V, gate, *a, **k):
        calls.append((tuple(V.shape), tuple(gate.shape)))
        return real(V, gate, *a, **k)

    monkeypatch.setattr(sp, "spectral_mix", counting)
    x = torch.from_numpy(d["x"]).to("cuda:0")
    with torch.no_grad():
        mh(x)
    assert len(calls) == 1
    (vs, gs), = calls
    assert vs[2] == mh.num_heads * mh.head_dim and gs[1] == mh.num_heads * mh.heads[0].G


def test_learnable_pos_phase_forces_the_autograd_path():
    """with frozen parameters and pos_phase.requires_grad the one-launch path would drop the gradient to pos_phase."""
    from fft_amd import SpectreMultiHead
    mh = SpectreMultiHead(16, 2, 32, pooling_type="mean", num_groups=2, wavelet_on_rate=0.0)
    for p in mh.parameters():
        p.requires_grad_(False)
    pp = torch.randn(17, dtype=torch.complex64, requires_grad=True)
    import fft_amd.spectre as sp
    calls = []
    for h in mh.heads:
        h.forward = (lambda *a, _h=h, **k: (calls.append("head"), (torch.zeros(a[0].shape), torch.zeros(a[0].shape[0], a[0].shape[2])))[1])   # stand-ins (mixed, q_pool): only the routing is under test
    real_apply = sp._SpectralMixFn.apply
    sp._SpectralMixFn.apply = staticmethod(lambda V, gate, mem, n_fft: (calls.append("node"), V * gate.abs().sum())[1])
    try:
        mh.out_proj = torch.nn.Identity()
        y = mh(torch.randn(1, 32, 16), pos_phase=pp)
        assert calls == ["node"] and y.requires_grad                                        # ONE autograd node for all heads, not the no-graph launch
        y.sum().backward()
        assert pp.grad is not None and pp.grad.abs().sum() > 0
        calls.clear()
        mh.fused_autograd = False
        mh(torch.randn(1, 32, 16), pos_phase=pp)
        assert calls == ["head", "head"]                                                    # the reference's per-head structure on request
    finally:
        sp._SpectralMixFn.apply = real_apply


def test_multihead_value_projection_node_equals_per_head_linears():
    """_MultiHeadValueFn (all heads' W_v into the channel slices of one tensor) against nn.Linear per chunk + cat, values and gradients."""
    from fft_amd.spectre import _MultiHeadValueFn
    torch.manual_seed(0)
    H, hd, B, N = 3, 8, 2, 5
    ws = [torch.randn(hd, hd, requires_grad=True) for _ in range(H)]
    x = torch.randn(B, N, H * hd, requires_grad=True)
    dout = torch.randn(B, N, H * hd)
    ref = torch.cat([c @ w.t() for c, w in zip(torch.chunk(x, H, dim=-1), ws)], dim=-1)
    gref = torch.autograd.grad((ref * dout).sum(), [x, *ws])
    got = _MultiHeadValueFn.apply(x, *ws)
    ggot = torch.autograd.grad((got * dout).sum(), [x, *ws])
    assert torch.allclose(got, ref, atol=1e-6)
    for a, b in zip(ggot, gref):
        assert torch.allclose(a, b, atol=1e-5)
    ws[1].requires_grad_(False)                                                             # a frozen head gets no weight gradient
    g2 = torch.autograd.grad((_MultiHeadValueFn.apply(x, *ws) * dout).sum(), [x, ws[0], ws[2]])
    assert torch.allclose(g2[0], gref[0], atol=1e-5) and torch.allclose(g2[2], gref[3], atol=1e-5)


def test_v_out_is_refused_under_autograd():
    from fft_amd import SpectreHead
    h = SpectreHead(8, 16, num_groups=2, pooling_type="mean")
    x = torch.randn(1, 16, 8)
    with pytest.raises(RuntimeError, match="inference only"):
        h.spectral_gate(x, v_out=torch.empty(1, 16, 8))
    with torch.no_grad():
        h.spectral_gate(x, v_out=torch.empty(1, 16, 8))
