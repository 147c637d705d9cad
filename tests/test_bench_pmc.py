"""bench.py's live `roofline.traffic` (two rocprofv3 --pmc child passes): the CSV arithmetic on the CPU tier, the collection itself on
the GPU tier.  The collection may be unavailable on a box (no rocprofv3, nested profiler): then bench.py replays profiles/pmc_latest.json
and says why, and the GPU test only checks that it says so."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _write_csv(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value\n")
        for i, (k, c, v) in enumerate(rows):
            f.write(f'{i},{i},1,"{k}",{c},{v}\n')


def test_parse_pmc_csv_picks_the_headline_kernel(tmp_path):
    import bench
    mix = "void sfft::spectre_mix_regtile64p<3, 3, false, false, false, true, true, 1>(sfft::RegtileArgs)"
    _write_csv(str(tmp_path / "host" / "123_counter_collection.csv"),
               [(mix, "FETCH_SIZE", 100.0), (mix, "FETCH_SIZE", 300.0), ("(anonymous namespace)::spectre_ticket_reset(unsigned int*, int)", "FETCH_SIZE", 7.0),
                (mix, "WRITE_SIZE", 50.0), ("void at::native::vectorized_elementwise_kernel<4>(int)", "FETCH_SIZE", 1e9)])
    vals, names = bench.parse_pmc_csv(str(tmp_path), "FETCH_SIZE")
    assert vals == [100.0, 300.0]
    assert names == ["void sfft::spectre_mix_regtile64p<3, 3, false, false, false, true, true, 1>"]
    assert bench.parse_pmc_csv(str(tmp_path), "WRITE_SIZE")[0] == [50.0]
    assert bench.parse_pmc_csv(str(tmp_path), "TCC_HIT_sum")[0] == []


def test_live_traffic_refuses_to_nest_and_can_be_switched_off(monkeypatch):
    import bench
    a = types.SimpleNamespace(io="f32", shape="2,4096,64", groups=2)
    monkeypatch.setenv("SPECTRE_BENCH_PMC", "0")
    assert bench.live_traffic(a) == (None, "SPECTRE_BENCH_PMC=0")
    monkeypatch.delenv("SPECTRE_BENCH_PMC")
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    rec, why = bench.live_traffic(a)
    assert rec is None and "nested" in why


@pytest.mark.gpu
def test_live_traffic_matches_the_algorithmic_bytes():
    """(64, 4096, 768) fp32: every line fetched once and written once -> FETCH_SIZE * 2 + WRITE_SIZE within a few per cent of the
    algorithmic bytes (profiles/r05_pmc_*: 1.01)."""
    import bench
    a = types.SimpleNamespace(io="f32", shape="64,4096,768", groups=4)
    rec, why = bench.live_traffic(a, timeout_s=240.0)
    if rec is None:
        pytest.skip(f"live PMC collection unavailable on this box: {why}")
    alg = bench.algorithmic_bytes(64, 4096, 4096, 768, 4, 4, 4)
    assert rec["launches"] == [bench.PMC_CHILD_LAUNCHES, bench.PMC_CHILD_LAUNCHES]
    assert all("spectre_mix_" in k for k in rec["kernels"])
    assert 0.97 < rec["hbm_bytes_per_launch"] / alg < 1.10, (rec, alg)
