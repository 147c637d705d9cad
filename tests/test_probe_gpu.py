"""spectre_probe_copy (the pure-copy probes bench.py reports as memory ceilings) and the allocation helper built on it.  Measurement
aids, not product path — but bench.py's ceilings are only as good as these copies are real copies."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    return torch.device("cuda:0")


@pytest.mark.parametrize("seg", [0, 128, 64, 32, -1, -2, -3])      # -1 / -2: flat float4 copy (plain / non-temporal), -3: hipMemcpyAsync
def test_copy_probe_copies_every_byte(seg):
    from fft_amd import copy_probe
    dev = _dev()
    B, N, D = 2, 4096, 256                                   # 8 MiB: whole 256-KiB chunks, whole 128-KiB tiles of 4096 rows
    src = torch.randn(B, N, D, device=dev)
    for per_cu in (1, 2, 4):
        dst = torch.full_like(src, float("nan"))
        ms = copy_probe(src, dst, seg, wgs_per_cu=per_cu, warmup=0, iters=1)
        assert ms > 0
        assert torch.equal(src, dst), (seg, per_cu)


def test_copy_probe_load_and_store_modes_and_bad_arguments():
    from fft_amd import copy_probe
    dev = _dev()
    src = torch.randn(2, 4096, 256, device=dev)
    dst = torch.zeros_like(src)
    assert copy_probe(src, dst, 128, mode="load", warmup=0, iters=1) > 0
    assert torch.count_nonzero(dst) == 0                     # load-only: nothing is written
    assert copy_probe(src, dst, 0, mode="store", warmup=0, iters=1) > 0
    with pytest.raises(ValueError):
        copy_probe(src, dst[:, :100], 0)                     # shapes differ
    with pytest.raises(ValueError):
        copy_probe(src, dst, 48)                             # segment size not a power of two (SPECTRE_E_INVALID)
    with pytest.raises(ValueError):
        copy_probe(src, dst, 64, tile_rows=1000)             # tile_rows does not divide the rows
    with pytest.raises(ValueError):
        copy_probe(src, dst, -1, mode="load")                # the flat forms are copies only
    with pytest.raises(ValueError):
        copy_probe(src, dst, -4)                             # no such form


def test_empty_on_fast_allocation():
    from fft_amd import empty_on_fast_allocation
    dev = _dev()
    t, ms = empty_on_fast_allocation((4, 4096, 256), torch.float32, dev, candidates=3)
    assert t.shape == (4, 4096, 256) and t.dtype == torch.float32 and t.is_cuda and len(ms) == 3 and all(m > 0 for m in ms)
    t2, ms2 = empty_on_fast_allocation((3, 5, 7), torch.float32, dev)      # not probe-able: plain torch.empty
    assert t2.shape == (3, 5, 7) and ms2 == []
