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
    with pytest.warns(UserWarning, match="WaveletRefinement"):                          # the default constructor works but says what it leaves out
        assert SpectreMultiHead(32, 2, 64).wavelet_refinement.on_rate == 0.0             # (reference default 0.1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        SpectreMultiHead(32, 2, 64, wavelet_on_rate=0.0)                                 # acknowledged: silent
    with pytest.raises(NotImplementedError, match="wavelet_on_rate"):
        SpectreMultiHead(32, 2, 64, wavelet_on_rate=0.1)                                # asking for the refinement is refused loudly
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
    y_graph = mh(x, pos_phase=pp, memory_fft=mem)                   # autograd path (per-head modules + cat)
    torch.cuda.synchronize()
    assert tuple(y.shape) == d["out"].shape
    assert_close(y.cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="multi-head forward")
    assert_close(y_graph.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="multi-head forward (autograd path)")
    y_graph.sum().backward()
    assert all(p.grad is not None for n, p in mh.named_parameters() if not n.startswith("wavelet_refinement"))


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
    calls = []
    for h in mh.heads:
        h.forward = (lambda *a, _h=h, **k: (calls.append(1), torch.zeros(a[0].shape))[1])   # stand-in: only the routing is under test
    mh.out_proj = torch.nn.Identity()
    mh(torch.randn(1, 32, 16), pos_phase=pp)
    assert len(calls) == 2                                                                  # per-head modules (graph path), not the fused launch


def test_v_out_is_refused_under_autograd():
    from fft_amd import SpectreHead
    h = SpectreHead(8, 16, num_groups=2, pooling_type="mean")
    x = torch.randn(1, 16, 8)
    with pytest.raises(RuntimeError, match="inference only"):
        h.spectral_gate(x, v_out=torch.empty(1, 16, 8))
    with torch.no_grad():
        h.spectral_gate(x, v_out=torch.empty(1, 16, 8))
