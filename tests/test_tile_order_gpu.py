"""Tile order of the persistent kernels as a product feature (round 6):
`spectre_plan_set_tile_order` / `get` (include/spectre_hip.h), the decision per shape CLASS (no pointers: rotating `out` buffers settle),
the optional per-pair mode with its LRU (more live pairs than entries, events still pending when an entry is evicted), two threads on
one plan while it measures, and the slice hand-out: one slice per stream, one per captured launch, the static map when none is left —
a stream stalled behind an event, a busy neighbour stream and a graph replay never share a slice.  Same bits in every case."""
import os
import subprocess
import sys
import threading

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


def _order(desc):
    return desc.rsplit("order=", 1)[1]


@pytest.fixture(autouse=True)
def _default_policy_afterwards():
    yield
    from fft_amd import set_tile_order
    for n in (4096, 3000):
        set_tile_order(n, "auto")


def test_set_and_get_tile_order_and_describe():
    from fft_amd import describe, get_tile_order, set_tile_order, spectral_mix
    V, gate = _problem(16, 4096, 96, 2)
    want = None
    for order in ("static", "tickets", "auto", "pair"):
        set_tile_order(4096, order)
        assert get_tile_order(4096) == order
        d = _order(describe(V, gate, None, 4096))
        assert d == order, d                                    # auto / pair: nothing measured yet
        y = spectral_mix(V, gate, None, 4096)
        torch.cuda.synchronize()
        want = y if want is None else want
        assert torch.equal(y, want), order                      # same bits whichever order
    with pytest.raises(ValueError):
        set_tile_order(4096, "fastest")
    V3, g3 = _problem(8, 3000, 64, 2)                          # the persistent mixed-radix kernels take the same switch
    set_tile_order(3000, "static")
    assert _order(describe(V3, g3, None, 3000)) == "static"
    a = spectral_mix(V3, g3, None, 3000)
    set_tile_order(3000, "tickets")
    assert _order(describe(V3, g3, None, 3000)) == "tickets"
    assert torch.equal(a, spectral_mix(V3, g3, None, 3000))


def test_auto_decides_per_class_although_the_out_buffer_rotates():
    """Round 5 keyed the decision on the (V, out) pointers: an allocator that hands out a fresh `out` every call never left the exploring
    state.  Now: one decision per shape class, behind 24 + 16 launches whatever the pointers are, and the static map only with a margin."""
    from fft_amd import describe, set_tile_order, spectral_mix
    set_tile_order(4096, "auto")
    V, gate = _problem(64, 4096, 192, 4, seed=11)
    first = spectral_mix(V, gate, None, 4096).clone()           # launch 1 (its own fresh out)
    outs = [torch.empty_like(V) for _ in range(7)]
    seen = []
    for i in range(70):
        out = outs[i % 7]
        spectral_mix(V, gate, None, 4096, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out, first), i
        seen.append(_order(describe(V, gate, None, 4096, out=outs[(i + 3) % 7])))
    assert seen[0] == "auto" and seen[-1].startswith(("auto:tickets", "auto:static")), seen[-1]
    final = seen[-1]
    if final.startswith("auto:static"):                         # "auto:static (A ms against B)": only with at least 1 % in hand
        a, b = float(final.split("(")[1].split(" ms")[0]), float(final.split("against ")[1].rstrip(")"))
        assert a < 0.99 * b, final
    k = next(i for i, s in enumerate(seen) if s == final)
    assert k <= 45, (k, seen)                                   # 24 warm + 16 timed (+ the launches it takes to see the last events)
    # another class (other dtype) starts from scratch, the first keeps its decision
    Vb = V.to(torch.bfloat16)
    assert _order(describe(Vb, gate, None, 4096)) == "auto"
    assert _order(describe(V, gate, None, 4096)) == final


def test_pair_mode_more_live_pairs_than_entries():
    """SPECTRE_ORDER_AUTO_PAIR: an LRU of 64 (V, out) pairs.  80 pairs, each launched past the warm-up so that event pairs are pending
    when its entry is evicted; nothing waits, nothing leaks into the results."""
    from fft_amd import describe, set_tile_order, spectral_mix
    set_tile_order(4096, "pair")
    V, gate = _problem(4, 4096, 32, 2, seed=5)
    want = spectral_mix(V, gate, None, 4096).clone()
    torch.cuda.synchronize()
    outs = [torch.empty_like(V) for _ in range(80)]
    for o in outs:
        for _ in range(30):                                     # 24 warm + 6 timed: pending events, undecided
            spectral_mix(V, gate, None, 4096, out=o)
    torch.cuda.synchronize()
    assert all(torch.equal(o, want) for o in outs)
    assert _order(describe(V, gate, None, 4096, out=outs[0])) == "pair"        # evicted long ago: as good as new
    assert _order(describe(V, gate, None, 4096, out=outs[-1])) == "pair"       # still exploring (30 of 40 launches)
    for _ in range(40):                                          # one pair all the way: settles
        spectral_mix(V, gate, None, 4096, out=outs[-1])
        torch.cuda.synchronize()
    assert _order(describe(V, gate, None, 4096, out=outs[-1])).startswith(("pair:tickets", "pair:static"))
    assert torch.equal(outs[-1], want)


def test_two_threads_on_one_plan_while_it_measures():
    from fft_amd import describe, set_tile_order, spectral_mix
    set_tile_order(4096, "auto")
    V, gate = _problem(8, 4096, 64, 2, seed=21)
    want = spectral_mix(V, gate, None, 4096).clone()
    torch.cuda.synchronize()
    errs = []

    def worker(seed):
        try:
            s = torch.cuda.Stream()
            outs = []
            with torch.cuda.stream(s):
                for _ in range(60):
                    outs.append(spectral_mix(V, gate, None, 4096))
            s.synchronize()
            if not all(torch.equal(o, want) for o in outs):
                errs.append(f"thread {seed}: wrong bits")
        except Exception as e:                                   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    spectral_mix(V, gate, None, 4096)
    torch.cuda.synchronize()
    assert _order(describe(V, gate, None, 4096)).startswith(("auto:tickets", "auto:static", "auto"))


def test_stalled_stream_busy_stream_and_graph_replay_never_share_a_slice():
    """with round 5's round-robin ring a stream that stalled behind an event while another issued 64 launches, or a
    graph replay beside eager launches, could end up on a slice still in use (its reset in the middle of the other kernel: missing tiles).
    Now a slice belongs to one stream, and a captured launch owns one for good."""
    from fft_amd import set_tile_order, spectral_mix
    set_tile_order(4096, "tickets")
    Va, ga = _problem(96, 4096, 128, 2, seed=31)                # ~0.2 ms a launch
    Vb, gb = _problem(4, 4096, 32, 2, seed=32)                  # short launches
    Vg, gg = _problem(48, 4096, 64, 2, seed=33)
    want_a, want_b, want_g = spectral_mix(Va, ga, None, 4096), spectral_mix(Vb, gb, None, 4096), spectral_mix(Vg, gg, None, 4096)
    out_g = torch.empty_like(Vg)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        spectral_mix(Vg, gg, None, 4096, out=out_g)
    sa, sb, sc, sg = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        gate_ev = torch.cuda.Event()
        with torch.cuda.stream(sc):
            torch.cuda._sleep(int(2.0e9 * 0.02))                 # ~ 10-20 ms: stream A's launches sit behind this
            gate_ev.record()
        outs_a, outs_b = [], []
        with torch.cuda.stream(sa):
            sa.wait_event(gate_ev)
            for _ in range(6):
                outs_a.append(spectral_mix(Va, ga, None, 4096))
        with torch.cuda.stream(sg):
            out_g.zero_()                                        # (on the replay's stream: ordered in front of it)
            sg.wait_event(gate_ev)
            graph.replay()
        with torch.cuda.stream(sb):                              # far more launches than there are slices, before / while A and the graph run
            for _ in range(400):
                outs_b.append(spectral_mix(Vb, gb, None, 4096))
        torch.cuda.synchronize()
        assert all(torch.equal(o, want_a) for o in outs_a), rep
        assert all(torch.equal(o, want_b) for o in outs_b), rep
        assert torch.equal(out_g, want_g), rep


def test_no_slice_left_means_the_static_map():
    """Every captured launch keeps a slice; a plan has 128.  The 129th and later ones (and streams that come after) take the static map:
    same bits.  (Own process: the plan is process-wide and stays exhausted.)"""
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from fft_amd import spectral_mix, set_tile_order\n"
            "g = torch.Generator().manual_seed(9)\n"
            "V = torch.randn(6, 4096, 64, generator=g).cuda(); gate = (torch.complex(torch.randn(6, 2, 2049, generator=g), torch.randn(6, 2, 2049, generator=g)) * 0.3).to(torch.complex64).cuda()\n"
            "set_tile_order(4096, 'tickets')\n"
            "want = spectral_mix(V, gate, None, 4096).clone(); torch.cuda.synchronize()\n"
            "outs = [torch.empty_like(V) for _ in range(140)]\n"
            "gr = torch.cuda.CUDAGraph()\n"
            "with torch.cuda.graph(gr):\n"
            "    for o in outs: spectral_mix(V, gate, None, 4096, out=o)\n"
            "for rep in range(2):\n"
            "    for o in outs: o.zero_()\n"
            "    gr.replay(); torch.cuda.synchronize()\n"
            "    assert all(torch.equal(o, want) for o in outs), rep\n"
            "s = torch.cuda.Stream()\n"
            "with torch.cuda.stream(s): y = spectral_mix(V, gate, None, 4096)\n"
            "torch.cuda.synchronize(); assert torch.equal(y, want)\n"
            "print('exhausted ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ))
    assert r.returncode == 0 and "exhausted ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
