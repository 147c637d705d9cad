"""Row N2 on a real GPU: the fused gate-producer tail (cubic resample -> modReLU -> positional phase, one HIP launch
through the C ABI) against (1) the gates the reference itself produced (fixtures: state_dict + x -> gate) and (2) the
PyTorch ops it replaces, at the benchmark size."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.spectral_mix_oracle import assert_close
from test_module_cpu import MODULE_CASES, _build

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _anchors_cpu(head, x):
    """Everything up to the gate MLP on the CPU with the same ops as the reference (spectre.py:502-516)."""
    with torch.no_grad():
        q_pool = head.q_norm(head.pooling(head.W_q(x)))
        anchors = torch.view_as_complex(head.gate_mlp(q_pool).view(x.shape[0], head.G, head.B, 2).contiguous())
        if head.use_toeplitz:                                                            # spectre.py:519-521 (fixtures g14_*)
            from fft_amd import complex_conv1d
            anchors = anchors + complex_conv1d(anchors, head.toeplitz_kernel, head.toeplitz_bw)
        return anchors


@pytest.mark.parametrize("path,cid", MODULE_CASES, ids=[c[1] for c in MODULE_CASES])
def test_fused_gate_matches_reference_gate(path, cid):
    from fft_amd import spectral_gate_fused
    d = load_golden(path)
    head = _build(d)
    x = torch.from_numpy(d["x"])
    pp = torch.from_numpy(d["pos_phase"]).to(DEV) if "pos_phase" in d else None
    anchors = _anchors_cpu(head, x).to(DEV)
    gate = spectral_gate_fused(anchors, head.modrelu.bias.detach().to(DEV), head.modrelu.eps_value, head.F_half, pp)
    torch.cuda.synchronize()
    ref = d["gate"]
    assert tuple(gate.shape) == ref.shape
    err = np.abs(gate.cpu().numpy() - ref)
    scale = np.sqrt(np.mean(np.abs(ref) ** 2))
    assert err.max() <= 2e-6 * scale + 1e-7, (cid, err.max(), scale)
    assert np.array_equal(gate.cpu().numpy() == 0, ref == 0)         # modReLU's exact zeros are the same bins


@pytest.mark.parametrize("shape", [(256, 4, 45, 2049), (3, 2, 4, 17), (2, 3, 38, 1501), (5, 1, 7, 8), (2, 2, 4, 2)],
                         ids=lambda s: "B%d_G%d_K%d_F%d" % s)
@pytest.mark.parametrize("phase", ["none", "F", "1F", "BF"])
def test_fused_gate_matches_the_ops_it_replaces(shape, phase):
    from fft_amd import ComplexModReLU, resample_complex, spectral_gate_fused
    B, G, K, F = shape
    g = torch.Generator().manual_seed(B * 1000 + F)
    anchors = (torch.complex(torch.randn(B, G, K, generator=g), torch.randn(B, G, K, generator=g)) * 0.5).to(DEV)
    mod = ComplexModReLU(G * F).to(DEV)
    with torch.no_grad():
        mod.bias.copy_(torch.randn(G * F, generator=g).to(DEV) * 0.3 - 0.1)
    pp = None
    if phase != "none":
        ang = torch.rand({"F": (F,), "1F": (1, F), "BF": (B, F)}[phase], generator=g) * 6.28
        pp = torch.polar(torch.ones_like(ang), ang).to(DEV)
    with torch.no_grad():
        ref = resample_complex(anchors, F, mode="cubic")
        ref = mod(ref.reshape(B, -1)).view_as(ref)
        if pp is not None:
            ref = ref * pp.unsqueeze(1 if pp.dim() == 2 else 0)
    got = spectral_gate_fused(anchors, mod.bias.detach(), mod.eps_value, F, pp)
    torch.cuda.synchronize()
    # the sample coordinate ((x + 1) / 2) * (K - 1) carries ~K ulp of rounding that differs between two correct float32
    # evaluations (fused multiply-add or not), and the cubic weights turn it into ~1e-5 of the anchor spacing
    scale = float(ref.abs().square().mean().sqrt())
    assert float((got - ref).abs().max()) <= 4e-5 * scale + 1e-7


def test_module_uses_the_fused_gate_in_inference_and_torch_ops_under_autograd():
    from fft_amd import SpectreHead
    torch.manual_seed(0)
    head = SpectreHead(64, 1024, num_groups=4, pooling_type="mean").to(DEV).eval()
    x = torch.randn(3, 1024, 64, device=DEV)
    pp = torch.polar(torch.ones(513), torch.rand(513) * 6.28).to(DEV)
    with torch.no_grad():
        _, g_fused, _ = head.spectral_gate(x, pp)
    _, g_ops, _ = head.spectral_gate(x, pp)               # grad enabled: the ops autograd can differentiate
    assert g_ops.requires_grad and not g_fused.requires_grad
    assert float((g_fused - g_ops.detach()).abs().max()) <= 4e-5 * float(g_ops.detach().abs().max())


def test_bad_arguments_fail_loudly():
    from fft_amd import spectral_gate_fused
    a = torch.zeros(2, 2, 4, dtype=torch.complex64, device=DEV)
    b = torch.zeros(2 * 9, device=DEV)
    with pytest.raises(RuntimeError):
        spectral_gate_fused(a.cpu(), b.cpu(), 1e-4, 9)
    with pytest.raises(ValueError):
        spectral_gate_fused(a, b[:5], 1e-4, 9)
    with pytest.raises(ValueError):
        spectral_gate_fused(a, b, 1e-4, 9, torch.zeros(3, 9, dtype=torch.complex64, device=DEV))


# ------------------------------------------------------------------------------------------------------
# backward of the fused tail (row N2 under autograd): against autograd through the reference's own ops
# (grid_sample bicubic / abs / relu / sqrt / mul — fft_amd.spectre.resample_complex + ComplexModReLU restate spectre.py:38-61, :109-121)
# ------------------------------------------------------------------------------------------------------
def _ops_gate(anchors, bias, eps, F_, phase):
    from fft_amd.spectre import resample_complex
    gate = resample_complex(anchors, F_, mode="cubic")
    B = anchors.shape[0]
    z = gate.reshape(B, -1)
    mag = torch.abs(z)
    z = z * (torch.relu(mag + bias) / torch.sqrt(mag.square() + eps * eps))
    gate = z.view_as(gate)
    if phase is not None:
        gate = gate * phase.unsqueeze(1 if phase.dim() == 2 else 0)
    return gate


@pytest.mark.parametrize("B,G,K,F_,ph", [(3, 2, 8, 129, None), (2, 4, 45, 2049, "shared"), (1, 1, 4, 17, "batch"), (5, 3, 11, 513, "batch"),
                                         (2, 2, 4, 33, "shared"), (4, 4, 64, 1025, None)])
def test_fused_gate_backward_matches_autograd_through_the_reference_ops(B, G, K, F_, ph):
    from fft_amd.spectre import _SpectralGateFn
    g = torch.Generator().manual_seed(B * 100 + K)
    anchors = torch.complex(torch.randn(B, G, K, generator=g), torch.randn(B, G, K, generator=g)).to("cuda:0")
    bias = (torch.randn(G * F_, generator=g) * 0.5 - 0.1).to("cuda:0")       # mixed signs: both sides of the relu
    phase = None
    if ph == "shared":
        phase = torch.exp(1j * torch.randn(F_, generator=g)).to(torch.complex64).to("cuda:0")
    elif ph == "batch":
        phase = torch.exp(1j * torch.randn(B, F_, generator=g)).to(torch.complex64).to("cuda:0")
    up = torch.complex(torch.randn(B, G, F_, generator=g), torch.randn(B, G, F_, generator=g)).to("cuda:0")
    grads = []
    for fused in (True, False):
        a = anchors.clone().requires_grad_(True)
        b = bias.clone().requires_grad_(True)
        p = phase.clone().requires_grad_(True) if phase is not None else None
        out = _SpectralGateFn.apply(a, b, p, 1e-4, F_) if fused else _ops_gate(a, b, 1e-4, F_, p)
        (out * up.conj()).real.sum().backward()                              # a generic real loss: every component gets its own weight
        grads.append((out.detach(), a.grad, b.grad, None if p is None else p.grad))
    torch.cuda.synchronize()
    (of, af, bf, pf), (oo, ao, bo, po) = grads
    assert_close(torch.view_as_real(of).cpu().numpy(), torch.view_as_real(oo).cpu().numpy(), rtol=1e-4, atol_rms=1e-5, what="gate")
    assert_close(torch.view_as_real(af).cpu().numpy(), torch.view_as_real(ao).cpu().numpy(), rtol=1e-4, atol_rms=2e-5, what="d anchors")
    assert_close(bf.cpu().numpy(), bo.cpu().numpy(), rtol=1e-4, atol_rms=2e-5, what="d bias")
    if pf is not None:
        assert_close(torch.view_as_real(pf).cpu().numpy(), torch.view_as_real(po).cpu().numpy(), rtol=1e-4, atol_rms=2e-5, what="d phase")


def test_module_training_gradients_agree_with_and_without_the_fused_gate_node():
    import copy
    from fft_amd import SpectreHead
    torch.manual_seed(3)
    head = SpectreHead(32, 512, num_groups=2, pooling_type="mean").to("cuda:0")
    ref = copy.deepcopy(head)
    ref.fused_gate_autograd = False
    x = torch.randn(3, 512, 32, device="cuda:0")
    w = torch.randn(3, 512, 32, device="cuda:0")
    for m in (head, ref):
        (m(x) * w).sum().backward()
    torch.cuda.synchronize()
    for (n, p), (_, q) in zip(head.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and q.grad is not None, n
        assert_close(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=2e-4, atol_rms=5e-5, what=n)
