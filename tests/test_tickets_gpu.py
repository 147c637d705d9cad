"""Dynamic tile order of the persistent n_fft = 4096 kernel (kernel_regtile64p.h TICKETS, round 5): one ticket per gang of workgroups
from a chip-wide counter.  The parity suites run through it (it is the default for fp32 rows); this module adds what only the ticket
protocol needs: the cases its claim bits and its sweep exist for (tools/tickets_lab: nobody gets a ticket, half the stream missing,
poisoned mailboxes, partial last gang — all bit-equal to the static map), the static map on request, many launches back to back on
two streams (every launch its own slice of the ring), and a captured graph (the slice reset is a node of the graph)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(B, N, D, G, seed=0):
    g = torch.Generator().manual_seed(seed)
    V = torch.randn(B, N, D, generator=g)
    gate = torch.complex(torch.randn(B, G, N // 2 + 1, generator=g), torch.randn(B, G, N // 2 + 1, generator=g)) * 0.3
    return V.to(DEV), gate.to(torch.complex64).to(DEV)


def test_ticket_protocol_cases_equal_the_static_map():
    exe = os.path.join(ROOT, "tools", "tickets_lab")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build_tools()
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    checks = [l for l in out.stdout.splitlines() if l.startswith("check ")]
    assert len(checks) >= 20, out.stdout[-3000:] + out.stderr[-2000:]
    bad = [l for l in checks if not l.rstrip().endswith("differing dwords: 0")]
    assert not bad and out.returncode == 0 and "all checks passed" in out.stdout, "\n".join(bad) + out.stderr[-2000:]


def test_describe_names_the_tile_order_and_the_order_is_measured_per_tensor_pair():
    """`order=auto` until a (V, out) pair has been measured (sixteen timed launches behind the first 24, T S S T ...), then `auto:tickets` or `auto:static`; the
    output is the same bit for bit whichever order a launch takes."""
    from fft_amd import describe, spectral_mix
    V, gate = _problem(64, 4096, 192, 4)
    out = torch.empty_like(V)
    assert describe(V, gate, None, 4096, out=out).endswith("order=auto")                              # fp32 rows, fast mode: not measured yet
    mem = torch.randn(2049, 192, dtype=torch.complex64, device=DEV)
    assert describe(V, gate, mem, 4096).endswith("order=auto")                                       # memory_fft: its own class (round 6: eligible too)
    first = spectral_mix(V, gate, None, 4096).clone()
    seen = set()
    for i in range(60):
        spectral_mix(V, gate, None, 4096, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, first), i                                                            # tickets or static: same bits
        seen.add(describe(V, gate, None, 4096, out=out).rsplit("order=", 1)[1].split(" (")[0])
    assert seen <= {"auto", "auto:tickets", "auto:static"} and (seen & {"auto:tickets", "auto:static"}), seen


def test_static_map_on_request_gives_the_same_bits():
    """SPECTRE_TUNING=1 SPECTRE_P64_TICKETS=0 (read once per process): the static tile map; both orders must produce identical bits."""
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from fft_amd import spectral_mix, describe\n"
            "g = torch.Generator().manual_seed(5)\n"
            "V = torch.randn(37, 4096, 208, generator=g).cuda(); gate = (torch.complex(torch.randn(37, 13, 2049, generator=g), torch.randn(37, 13, 2049, generator=g)) * 0.3).to(torch.complex64).cuda()\n"
            "print(describe(V, gate, None, 4096))\n"
            "y = spectral_mix(V, gate, None, 4096); torch.cuda.synchronize()\n"
            "torch.save(y.cpu(), sys.argv[1])\n" % ROOT)
    outs = []
    for env_extra, want in (({"SPECTRE_TUNING": "1", "SPECTRE_TILE_ORDER": "tickets"}, "order=tickets"), ({"SPECTRE_TUNING": "1", "SPECTRE_TILE_ORDER": "static"}, "order=static")):
        path = os.path.join(ROOT, "gpurun_out", f"tickets_ab_{len(outs)}.pt")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env_extra))
        assert r.returncode == 0 and want in r.stdout, r.stdout + r.stderr[-2000:]
        outs.append(torch.load(path))
        os.remove(path)
    assert torch.equal(outs[0], outs[1])


def test_many_launches_on_two_streams_each_with_its_own_slice():
    from fft_amd import spectral_mix
    V1, g1 = _problem(24, 4096, 96, 2, seed=1)
    V2, g2 = _problem(16, 4096, 160, 2, seed=2)
    want1, want2 = spectral_mix(V1, g1, None, 4096), spectral_mix(V2, g2, None, 4096)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs1, outs2 = [], []
    for _ in range(40):                                   # more launches than the ring has slices, interleaved on two streams
        with torch.cuda.stream(s1):
            outs1.append(spectral_mix(V1, g1, None, 4096))
        with torch.cuda.stream(s2):
            outs2.append(spectral_mix(V2, g2, None, 4096))
    torch.cuda.synchronize()
    assert all(torch.equal(o, want1) for o in outs1) and all(torch.equal(o, want2) for o in outs2)


def test_captured_graph_resets_its_slice_on_every_replay():
    from fft_amd import describe, spectral_mix
    V, gate = _problem(12, 4096, 64, 4, seed=3)
    assert "order=auto" in describe(V, gate, None, 4096)
    out = torch.empty_like(V)
    spectral_mix(V, gate, None, 4096, out=out)            # eager once: plan + LDS opt-in
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        spectral_mix(V, gate, None, 4096, out=out)
    for seed in (7, 8, 9):
        V.copy_(torch.randn(V.shape, generator=torch.Generator().manual_seed(seed)).to(DEV))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, spectral_mix(V, gate, None, 4096))
