"""Row N4 on a real GPU: prefill (spectre_rfft_fwd) and decode (spectre_decode_step + spectre_gate_fwd) through the C ABI
against what the reference produced (fixtures g10_decode_*) and against the CPU oracle at the benchmark width."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.decode_oracle import PrefixFFTCacheOracle, decode_step_oracle
from oracle.spectral_mix_oracle import assert_close
from test_decode_cpu import DECODE, IDS
from test_module_cpu import _build

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("path", DECODE, ids=IDS)
def test_decode_matches_reference_fixtures(path):
    from fft_amd import PrefixFFTCache
    d = load_golden(path)
    head = _build(d).to(DEV)
    n_fft, dim = int(d["n_fft"]), d["Vp"].shape[1]
    cache = PrefixFFTCache(n_fft, dim, device=torch.device(DEV))
    cache.prefill(torch.from_numpy(d["Qp"]).to(DEV), torch.from_numpy(d["Vp"]).to(DEV))
    ref = d["prefix_after_prefill"]
    assert_close(torch.view_as_real(cache.prefix_fft).cpu().numpy(), ref.view(np.float32).reshape(*ref.shape, 2), what="prefill")
    if "mem" in d:
        cache.prefix_fft += torch.from_numpy(d["mem"]).to(DEV)      # usage documented at spectre.py:736-741
    q, v = torch.from_numpy(d["q_seq"]).to(DEV), torch.from_numpy(d["v_seq"]).to(DEV)
    outs = torch.stack([head.decode_step(q[i], v[i], cache) for i in range(q.shape[0])])
    torch.cuda.synchronize()
    for i in range(q.shape[0]):                               # every step on its own scale
        assert_close(outs[i].cpu().numpy(), d["outs"][i], what=f"step {i}")
    pf = d["prefix_final"]
    assert_close(torch.view_as_real(cache.prefix_fft).cpu().numpy(), pf.view(np.float32).reshape(*pf.shape, 2), what="final spectrum")
    assert_close(cache.sum_q.cpu().numpy(), d["sum_q_final"], what="sum_q")
    assert cache.t == int(d["t_final"])


@pytest.mark.parametrize("shape", [(4096, 768, 4), (3000, 96, 2), (1024, 64, 4), (97, 10, 2)], ids=lambda s: "n%d_d%d_G%d" % s)
def test_decode_matches_oracle_at_size(shape):
    from fft_amd import PrefixFFTCache, SpectreHead
    n_fft, dim, G = shape
    torch.manual_seed(n_fft)
    head = SpectreHead(dim, n_fft, num_groups=G, pooling_type="mean").eval()
    L, T = n_fft - 3, 8                                      # wraps after three steps
    Qp, Vp = torch.randn(L, dim), torch.randn(L, dim)
    q, v = torch.randn(T, dim), torch.randn(T, dim)
    oc = PrefixFFTCacheOracle(n_fft, dim)
    oc.prefill(Qp, Vp)
    want = torch.stack([decode_step_oracle(head, q[i], v[i], oc) for i in range(T)])
    head = head.to(DEV)
    cache = PrefixFFTCache(n_fft, dim, device=torch.device(DEV))
    cache.prefill(Qp.to(DEV), Vp.to(DEV))
    got = torch.stack([head.decode_step(q[i].to(DEV), v[i].to(DEV), cache) for i in range(T)])
    torch.cuda.synchronize()
    for i in range(T):
        assert_close(got[i].cpu().numpy(), want[i].numpy(), rtol=1e-4, atol_rms=2e-4, what=f"step {i}")
    assert_close(torch.view_as_real(cache.prefix_fft).cpu().numpy(), torch.view_as_real(oc.prefix_fft).numpy(), what="spectrum")


def test_state_only_decode_step_and_prefill_shapes():
    from fft_amd import PrefixFFTCache, rfft_prefill
    torch.manual_seed(3)
    V = torch.randn(3, 200, 24, device=DEV)
    for n in (256, 200, 128, 210):
        spec = rfft_prefill(V, n)
        ref = torch.fft.rfft(V.cpu().double(), n=n, dim=1)
        assert_close(torch.view_as_real(spec).cpu().numpy(), torch.view_as_real(ref).numpy(), what=f"rfft n={n}")
    spec = rfft_prefill(V[0, :, :7].bfloat16(), 256)         # (N, D) input, odd D, bf16 storage
    ref = torch.fft.rfft(V[0, :, :7].bfloat16().double().cpu(), n=256, dim=0)
    assert_close(torch.view_as_real(spec).cpu().numpy(), torch.view_as_real(ref).numpy(), what="rfft 2-D bf16")
    c = PrefixFFTCache(64, 8, device=torch.device(DEV))
    oc = PrefixFFTCacheOracle(64, 8)
    Q, Vp = torch.randn(64, 8), torch.randn(64, 8)
    c.prefill(Q.to(DEV), Vp.to(DEV))
    oc.prefill(Q, Vp)
    for i in range(5):
        qt, vt = torch.randn(8), torch.randn(8)
        pf, sq = c.decode_step(qt.to(DEV), vt.to(DEV))
        opf, osq = oc.decode_step(qt, vt)
    assert pf is c.prefix_fft
    assert_close(torch.view_as_real(pf).cpu().numpy(), torch.view_as_real(opf).numpy(), what="state-only spectrum")
    assert_close(sq.cpu().numpy(), osq.numpy(), what="sum_q")
    assert torch.equal(c.V_buf.cpu(), oc.V_buf) and torch.equal(c.Q_buf.cpu(), oc.Q_buf)


def test_fused_and_ops_decode_paths_agree():
    """SpectreHead.decode_step = one C-ABI call; the ops path (kept for non-standard gate MLPs) must give the same step."""
    from fft_amd import PrefixFFTCache, SpectreHead
    from fft_amd.decode import _head_decode_step_ops
    torch.manual_seed(5)
    n_fft, dim = 60, 48
    head = SpectreHead(dim, n_fft, num_groups=2, pooling_type="mean").to(DEV).eval()
    Qp, Vp = torch.randn(55, dim, device=DEV), torch.randn(55, dim, device=DEV)
    q, v = torch.randn(20, dim, device=DEV), torch.randn(20, dim, device=DEV)
    c1 = PrefixFFTCache(n_fft, dim, device=torch.device(DEV)); c1.prefill(Qp, Vp)
    c2 = PrefixFFTCache(n_fft, dim, device=torch.device(DEV)); c2.prefill(Qp, Vp)
    for i in range(20):
        a = head.decode_step(q[i], v[i], c1)
        b = _head_decode_step_ops(head, q[i], v[i], c2)
        assert_close(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-4, atol_rms=1e-4, what=f"step {i}")
    assert c1.t == c2.t
    assert_close(c1.sum_q.cpu().numpy(), c2.sum_q.cpu().numpy(), what="sum_q")
    assert torch.equal(c1.V_buf, c2.V_buf) and torch.equal(c1.Q_buf, c2.Q_buf)


def test_fused_step_validates_its_operands():
    """the single-call decode step reads raw pointers as d floats — mismatched sizes must raise, a bf16 prompt must
    not turn the running query sum into a half-sized buffer."""
    import fft_amd
    d, N = 32, 256
    head = fft_amd.SpectreHead(d, N, num_groups=2, pooling_type="mean").to("cuda:0").eval()
    cache = fft_amd.PrefixFFTCache(N, d, device="cuda:0")
    Q = torch.randn(10, d, device="cuda:0")
    V = torch.randn(10, d, device="cuda:0")
    cache.prefill(Q.to(torch.bfloat16), V.to(torch.bfloat16))
    assert cache.sum_q.dtype == torch.float32 and cache.sum_q.numel() == d
    y = head.decode_step(torch.randn(d, device="cuda:0"), torch.randn(d, device="cuda:0"), cache)
    torch.cuda.synchronize()
    assert y.shape == (d,) and bool(torch.isfinite(y).all())
    with pytest.raises(ValueError):
        head.decode_step(torch.randn(d + 1, device="cuda:0"), torch.randn(d, device="cuda:0"), cache)
    with pytest.raises(ValueError):
        head.decode_step(torch.randn(d, device="cuda:0"), torch.randn(d), cache)                 # v_t on the CPU
    with pytest.raises(ValueError):
        head.decode_step(torch.randn(d, device="cuda:0"), torch.randn(d, device="cuda:0"), fft_amd.PrefixFFTCache(2 * N, d, device="cuda:0"))
    with pytest.raises(ValueError):
        head.decode_step(torch.randn(d, device="cuda:0"), torch.randn(d, device="cuda:0"), fft_amd.PrefixFFTCache(N, 2 * d, device="cuda:0"))
