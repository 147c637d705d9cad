"""The N>1 path on CPU: world_size-2 `gloo` processes shard the batch exactly as bench.py does on GPUs
(contiguous runs of B per rank, memory_fft replicated, NO collective in the data path), each rank runs the
mix on its shard (the oracle stands in for the kernel — there is no GPU here), and the concatenation of
the shards must equal the unsharded result; the timed region uses bench.py's own protocol (agreed start, node-clock stamps,
first start -> last finish)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, N, D, G, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fft_amd import batch_shard
    from oracle.spectral_mix_oracle import spectral_mix_torch
    gen = torch.Generator().manual_seed(123)                 # every rank builds the same global problem
    V = torch.randn(B, N, D, generator=gen)
    F = N // 2 + 1
    gate = torch.complex(torch.randn(B, G, F, generator=gen), torch.randn(B, G, F, generator=gen)) * 0.3
    mem = torch.complex(torch.randn(F, D, generator=gen), torch.randn(F, D, generator=gen)) * 0.1
    s, e = batch_shard(B, world, rank)
    y_local = spectral_mix_torch(V[s:e], gate[s:e], mem, N)  # shard of V and gate, replicated mem
    # bench.py's timing protocol (fft_amd/rendezvous.py): barrier, agreed start on the node clock, per-rank stamps, whole job =
    # first rank's start -> last rank's finish.  Rank r "works" for 20 ms * (r + 1).
    import time
    from fft_amd.rendezvous import Rendezvous
    rdv = Rendezvous(world=world, rank=rank, backend="gloo")
    rdv.barrier()
    t_go = rdv.common_start(margin_s=0.01)
    t_start = rdv.node_clock()
    assert 0.0 <= t_start - t_go < 0.005                     # every rank left the spin at the agreed instant
    time.sleep(0.02 * (rank + 1))
    t_end = rdv.node_clock()
    w = rdv.job_window(t_start, t_end)
    assert w["wall_s"] >= w["max_rank_wall_s"] >= 0.02 * world
    assert w["wall_s"] - w["max_rank_wall_s"] < 0.005         # common start: the job window is the slowest rank's, not more
    assert 0.0 <= w["start_after_first_us"] < 5000.0
    ends = rdv.gather_over_ranks(w["end_before_last_us"])
    assert min(ends) == 0.0 and (world == 1 or max(ends) > 15000.0)   # the last rank defines the end; rank 0 finished ~20 ms earlier
    # gather only to CHECK (outside any timed region); the data path itself needs no collective
    sizes = [batch_shard(B, world, r) for r in range(world)]
    outs = [torch.empty(se - ss, N, D) for ss, se in sizes]
    dist.all_gather(outs, y_local) if len({o.shape for o in outs}) == 1 else dist.all_gather_object(outs, y_local)
    if rank == 0:
        full = spectral_mix_torch(V, gate, mem, N)
        cat = torch.cat([torch.as_tensor(o) for o in outs], dim=0)
        np.save(os.path.join(tmp, "maxdiff.npy"), np.array([(cat - full).abs().max().item(), float(cat.shape[0])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_batch_shard_world2_gloo(tmp_path, B):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), B, 64, 8, 2, str(tmp_path)), nprocs=world, join=True)
    maxdiff, rows = np.load(tmp_path / "maxdiff.npy")
    assert rows == B
    assert maxdiff == 0.0        # per-(b, c) independence: sharding changes nothing, bit for bit


def _gpu_worker(rank, world, port, B, N, D, G, use_mem, tmp):
    """Same protocol as _worker, but every rank runs the PRODUCT (fft_amd.spectral_mix through the C ABI) on its shard; the ranks
    share cuda:0 on a 1-GPU box (one process per GPU on a node — bench.py's layout — when more are visible)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # no tensor data crosses ranks in the data path
    from fft_amd import batch_shard, spectral_mix
    dev = torch.device(f"cuda:{rank % torch.cuda.device_count()}")
    gen = torch.Generator().manual_seed(321)
    V = torch.randn(B, N, D, generator=gen)
    F = N // 2 + 1
    gate = (torch.complex(torch.randn(B, G, F, generator=gen), torch.randn(B, G, F, generator=gen)) * 0.3).to(torch.complex64)
    mem = (torch.complex(torch.randn(F, D, generator=gen), torch.randn(F, D, generator=gen)) * 0.1).to(torch.complex64) if use_mem else None
    mem_d = None if mem is None else mem.to(dev)                        # replicated on every rank
    s, e = batch_shard(B, world, rank)
    y_local = spectral_mix(V[s:e].to(dev), gate[s:e].to(dev), mem_d, N).cpu()
    outs = [None] * world
    dist.all_gather_object(outs, y_local)                              # check only, outside any timed region
    if rank == 0:
        full = spectral_mix(V.to(dev), gate.to(dev), mem_d, N).cpu()
        cat = torch.cat(outs, dim=0)
        from oracle.spectral_mix_oracle import spectral_mix_numpy
        ref = spectral_mix_numpy(V.numpy(), gate.numpy(), None if mem is None else mem.numpy(), N)
        rms = float(np.sqrt((ref ** 2).mean()))
        np.save(os.path.join(tmp, "gpu.npy"), np.array([(cat - full).abs().max().item(), float(cat.shape[0]),
                                                         float(np.abs(cat.numpy() - ref).max() / rms)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,D,G,use_mem", [(5, 4096, 64, 4, False), (6, 1024, 48, 3, True), (3, 3000, 32, 2, False)])
def test_batch_shard_world2_runs_the_hip_kernels(tmp_path, B, N, D, G, use_mem):
    world = 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), B, N, D, G, use_mem, str(tmp_path)), nprocs=world, join=True)
    maxdiff, rows, err = np.load(tmp_path / "gpu.npy")
    assert rows == B
    assert maxdiff == 0.0        # the kernels treat every (b, c) column independently: sharding changes nothing, bit for bit
    assert err < 2e-5            # and the sharded result is the reference's (fp64 oracle, relative to RMS)
