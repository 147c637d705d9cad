"""The drop-in module on a real GPU: reference state_dict + x -> the reference's output (fixtures)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.spectral_mix_oracle import assert_close
from test_module_cpu import MODULE_CASES, _build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path,cid", MODULE_CASES, ids=[c[1] for c in MODULE_CASES])
def test_module_forward_matches_reference(path, cid):
    d = load_golden(path)
    head = _build(d).to("cuda:0")
    x = torch.from_numpy(d["x"]).to("cuda:0")
    pp = torch.from_numpy(d["pos_phase"]).to("cuda:0") if "pos_phase" in d else None
    mem = torch.from_numpy(d["mem"]).to("cuda:0") if "mem" in d else None
    with torch.no_grad():
        out, q_pool = head(x, pos_phase=pp, return_q_pool=True, memory_fft=mem)
        out2 = head(x, pos_phase=pp, memory_fft=mem)
    torch.cuda.synchronize()
    assert tuple(out.shape) == d["out"].shape and q_pool.shape == (x.shape[0], x.shape[2])
    assert torch.equal(out, out2)                                   # deterministic in eval()
    # GEMMs and the gate producer run on the GPU here (hipBLASLt vs MKL): allow their fp32 noise too
    assert_close(out.cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what=cid)


def test_memory_fft_gradient_reaches_an_unfrozen_memory():
    """The reference freezes memory_fft (spectre.py:961) but autograd would train it if a caller un-freezes it: the layer must return that
    gradient (round 6; it used to refuse), with frozen parameters too (the routing then depends on the memory alone), and the same value the
    oracle's torch restatement gives for the head's own V and gate.  (Reference-made fixtures: tests/test_block.py `*_trainmem`.)"""
    from fft_amd import SpectreHead
    from oracle.spectral_mix_oracle import assert_close, spectral_mix_torch
    head = SpectreHead(32, 256, num_groups=2, pooling_type="mean").to("cuda:0")
    for p in head.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 256, 32, device="cuda:0")
    mem = torch.randn(129, 32, dtype=torch.complex64, device="cuda:0", requires_grad=True)
    out = head(x, memory_fft=mem)
    assert out.requires_grad
    dout = torch.randn_like(out)
    (out * dout).sum().backward()
    with torch.no_grad():
        V, gate, _ = head.spectral_gate(x)
    m = mem.detach().cpu().requires_grad_(True)
    (spectral_mix_torch(V.cpu(), gate.to(torch.complex64).cpu(), m, 256) * dout.cpu()).sum().backward()
    g, e = mem.grad.cpu().numpy(), m.grad.numpy()
    assert_close(np.stack((g.real, g.imag)), np.stack((e.real, e.imag)), rtol=1e-4, atol_rms=1e-4, what="d/d memory_fft")
    head(x, memory_fft=mem.detach()).sum()                            # frozen memory (as in the reference): no graph needed


def test_mean_pooling_fold_is_the_same_layer():
    """mean_n(W_q x) == W_q(mean_n x): folding the query GEMM into the pooled vector changes nothing but rounding."""
    from fft_amd import SpectreHead
    torch.manual_seed(1)
    head = SpectreHead(64, 512, num_groups=4, pooling_type="mean").to("cuda:0").eval()
    x = torch.randn(3, 512, 64, device="cuda:0")
    with torch.no_grad():
        y1, q1 = head(x, return_q_pool=True)
        head.fold_mean_pooling = False
        y0, q0 = head(x, return_q_pool=True)
    assert float((q1 - q0).abs().max()) <= 2e-5 * float(q0.abs().max())
    assert_close(y1.cpu().numpy(), y0.cpu().numpy(), rtol=1e-4, atol_rms=1e-4, what="folded vs unfolded")


def test_forward_can_be_captured_in_a_graph():
    """The library only enqueues kernels on the caller's stream: after one warm-up call (plan upload, LDS opt-in) a layer
    forward replays from a hipGraph with identical results ("Graph capture and streams")."""
    from fft_amd import SpectreHead
    torch.manual_seed(2)
    head = SpectreHead(64, 1024, num_groups=4, pooling_type="mean").to("cuda:0").eval()
    x = torch.randn(2, 1024, 64, device="cuda:0")
    with torch.no_grad():
        want = head(x)                                   # warm-up: builds the plan
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = head(x)
        x.copy_(torch.randn(2, 1024, 64, device="cuda:0"))
        g.replay()
        torch.cuda.synchronize()
        want2 = head(x)
    assert not torch.equal(want, want2)
    assert torch.equal(y, want2)


def test_modrelu_bias_and_pos_phase_get_gradients_when_the_rest_is_frozen():
    """with W_q / q_norm / gate_mlp frozen and x without grad, the anchors carry no graph — the fused gate launch must
    not be taken then, or modrelu.bias and a learnable pos_phase silently get no gradient (the reference trains them)."""
    from fft_amd import SpectreHead
    torch.manual_seed(0)
    head = SpectreHead(32, 256, num_groups=2, pooling_type="mean").to("cuda:0")
    for m in (head.W_q, head.q_norm, head.gate_mlp):
        for p in m.parameters():
            p.requires_grad_(False)
    x = torch.randn(2, 256, 32, device="cuda:0")
    pos = torch.nn.Parameter(torch.exp(1j * torch.randn(129, device="cuda:0")).to(torch.complex64))
    head(x, pos_phase=pos).square().sum().backward()
    torch.cuda.synchronize()
    assert head.modrelu.bias.grad is not None and float(head.modrelu.bias.grad.abs().sum()) > 0
    assert pos.grad is not None and float(pos.grad.abs().sum()) > 0
    assert head.W_v.weight.grad is not None


def test_toeplitz_head_matches_the_references_forward_and_autograd():
    """use_toeplitz=True (the option the reference's constructor cannot build): outputs and autograd of the reference's own forward run with the
    parameter attached by hand (fixture g14_toeplitz_bw2_grad, tests/golden/make_golden.py `case_toeplitz`) — d/dx and every parameter gradient,
    the complex Toeplitz kernel's included."""
    import os
    from conftest import GOLDEN_DIR
    d = load_golden(os.path.join(GOLDEN_DIR, "g14_toeplitz_bw2_grad.npz"))
    head = _build(d).to("cuda:0")
    assert head.use_toeplitz and head.toeplitz_kernel.numel() == 5
    x = torch.from_numpy(d["x"]).to("cuda:0").requires_grad_(True)
    out = head(x)
    (out * torch.from_numpy(d["dout"]).to("cuda:0")).sum().backward()
    torch.cuda.synchronize()
    assert_close(out.detach().cpu().numpy(), d["out"], rtol=1e-4, atol_rms=2e-4, what="forward")
    assert_close(x.grad.cpu().numpy(), d["grad_x"], rtol=1e-4, atol_rms=5e-4, what="d/dx")
    checked = 0
    for name, prm in head.named_parameters():
        g, e = prm.grad.cpu().numpy(), d["grad/" + name]
        if np.iscomplexobj(e):
            g, e = np.stack((g.real, g.imag)), np.stack((e.real, e.imag))
        assert_close(g, e, rtol=1e-4, atol_rms=1e-3, what="d/d " + name)
        checked += 1
    assert checked == 10 and "grad/toeplitz_kernel" in d
    # decode with the option: the single-call path has no Toeplitz step, so the step runs the anchors through PyTorch ops — and must agree with
    # the same step computed by hand from the head's own pieces
    from fft_amd import PrefixFFTCache, complex_conv1d, spectral_gate_fused
    head = head.eval()
    cache = PrefixFFTCache(head.n_fft, head.d, device="cuda:0")
    with torch.no_grad():
        prompt = torch.randn(20, head.d, device="cuda:0")
        cache.prefill(head.W_q(prompt), head.W_v(prompt))
        y = head.decode_step(head.W_q(prompt[3]), head.W_v(prompt[4]), cache)
    assert y.shape == (head.d,) and torch.isfinite(y).all()
