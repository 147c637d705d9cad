"""Rank rendezvous for the batch-sharded spectral mix (shard B, no data-path collective).

The data path never talks between ranks; a multi-GPU run only needs three control-plane operations — a barrier on both sides of the timed
region, a MAX over ranks of two floats, and a census of which rank drives which device.  `rendezvous()` provides them so that a first run on
an 8-GPU node cannot be lost to the transport:

  1. the process group is ALWAYS formed over gloo first (TCP on 127.0.0.1: no GPU, no RCCL involved) and the census is gathered there:
     `ranks_seen` = one record per rank (rank, local rank, device ordinal, device UUID, pid);
  2. if RCCL is wanted (`prefer="nccl"`), a HIP device is present and every rank drives a DISTINCT device, an RCCL group is created next
     to it and proved with one barrier and one MAX all-reduce of a device tensor, inside a time box (ours: the RCCL watchdog is disarmed
     for the proof, so a hung collective cannot abort the process before the fallback); the ranks then AGREE over gloo whether
     it worked everywhere.  Any exception, any time-out on any rank, or two ranks on one device (RCCL refuses that: "duplicate GPU")
     -> every rank uses gloo for the barrier / MAX and `fallback_reason` says why;
  3. `barrier()` and `max_over_ranks()` use whichever transport was agreed on; `backend` names it ("nccl" | "gloo" | "none" for one rank).

Everything here runs without a GPU (the CPU-tier test drives a world of 2 over gloo); bench.py and the multi-GPU tests share it.
"""
from __future__ import annotations

import datetime
import socket
import os
import threading
from dataclasses import dataclass, field
from typing import Callable, List, Optional


@dataclass
class Rendezvous:
    world: int
    rank: int
    backend: str                                   # "none" (single rank) | "nccl" (= RCCL on ROCm) | "gloo"
    ranks_seen: List[dict] = field(default_factory=list)
    fallback_reason: Optional[str] = None          # why RCCL was wanted but gloo is used (None if not applicable)
    oversubscribed: bool = False                   # several ranks share one device
    _nccl_group: object = None
    _device: object = None

    def barrier(self) -> None:
        """All ranks arrive (the caller adds torch.cuda.synchronize() around it for the timed region)."""
        if self.backend == "none":
            return
        import torch.distributed as dist
        if self.backend == "nccl":
            dist.barrier(group=self._nccl_group, device_ids=[self._device.index])
        else:
            dist.barrier()

    def max_over_ranks(self, values: List[float]) -> List[float]:
        if self.backend == "none":
            return [float(v) for v in values]
        import torch
        import torch.distributed as dist
        if self.backend == "nccl":
            t = torch.tensor(values, dtype=torch.float64, device=self._device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._nccl_group)
        else:
            t = torch.tensor(values, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    @staticmethod
    def node_clock() -> float:
        """CLOCK_MONOTONIC: one clock for every process of the node (the ranks of a single-node job can compare stamps directly)."""
        import time
        return time.clock_gettime(time.CLOCK_MONOTONIC)

    def common_start(self, margin_s: float = 0.004) -> float:
        """The ranks leave a gloo barrier up to a millisecond apart (TCP on the loopback); on a 26-ms timed region that reads as
        scaling loss.  Agree on a start time `margin_s` ahead of the LATEST rank (the MAX all-reduce is itself a barrier) and spin
        until the node clock reaches it.  Returns the agreed time; one rank: returns at once."""
        if self.backend == "none":
            return self.node_clock()
        (t_go,) = self.max_over_ranks([self.node_clock() + margin_s])
        while self.node_clock() < t_go:
            pass
        return t_go

    def job_window(self, t_start: float, t_end: float) -> dict:
        """Whole-job wall time from per-rank node-clock stamps: first rank's start -> last rank's finish (never less than the MAX of the
        per-rank durations, the contract's figure, and it also counts a rank that started late).  Same dict on every rank except the
        two offsets, which are this rank's own."""
        last_end, neg_first_start, longest = self.max_over_ranks([t_end, -t_start, t_end - t_start])
        # CLOCK_MONOTONIC is one clock for the processes of ONE node only: across nodes the stamps are not comparable and the window is the
        # contract's MAX over ranks of the per-rank duration
        one_node = len({r.get("host") for r in self.ranks_seen}) <= 1
        return {"wall_s": (last_end + neg_first_start) if one_node else longest, "max_rank_wall_s": longest, "single_node": one_node,
                "start_after_first_us": (t_start + neg_first_start) * 1e6, "end_before_last_us": (last_end - t_end) * 1e6}

    def gather_over_ranks(self, obj) -> list:
        """[obj of rank 0, obj of rank 1, ...] on every rank — always over the gloo control plane (small Python objects: per-rank timings
        for the attribution fields of the bench line), whatever transport the barrier uses."""
        if self.backend == "none":
            return [obj]
        import torch.distributed as dist
        got = [None] * self.world
        dist.all_gather_object(got, obj)             # the DEFAULT group = gloo (rendezvous() forms it first; the RCCL group sits beside it)
        return got

    def describe(self) -> dict:
        return {"rendezvous": self.backend, "ranks_seen": self.ranks_seen, "distinct_devices": len({(r["device"], r["uuid"]) for r in self.ranks_seen}),
                "oversubscribed": self.oversubscribed, "rendezvous_fallback": self.fallback_reason}

    def close(self) -> None:
        if self.backend != "none":
            import torch.distributed as dist
            try:
                dist.barrier()                      # gloo: nobody tears the store down while a peer still reads it
                dist.destroy_process_group()
            except Exception:
                pass


def _device_identity(device) -> dict:
    """What this rank drives: ordinal + UUID (two ranks that see the same physical GPU under different ordinals still collide)."""
    ident = {"device": None, "uuid": None}
    if device is None:
        return ident
    import torch
    ident["device"] = int(device.index)
    try:
        ident["uuid"] = str(torch.cuda.get_device_properties(device).uuid)
    except Exception:
        ident["uuid"] = None
    return ident


def _prove_nccl(device, timeout_s: float):
    """Create an RCCL group beside the gloo one and push one barrier + one MAX all-reduce through it (raises on any failure)."""
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(device)                   # (the current device is per thread, and rendezvous() calls this from a worker thread)
    # The time box is OURS (the caller joins this thread for timeout_s and then agrees with the other ranks over gloo).  ProcessGroupNCCL's
    # own watchdog must not get there first: with the default TORCH_NCCL_ASYNC_ERROR_HANDLING a collective that exceeds the GROUP's timeout
    # makes the watchdog abort the whole PROCESS (SIGABRT) — before the "agree and fall back" step could run.  So: no
    # process-level handling (read when the group is constructed), no heartbeat monitor, and a group timeout far beyond the box.
    # The variables are read when the group is CONSTRUCTED, so they are set around new_group() only and put back afterwards:
    # NCCL groups the host application creates later keep their watchdog.  (A caller-set value is respected: setdefault semantics.)
    disarm = {"TORCH_NCCL_ASYNC_ERROR_HANDLING": "0", "TORCH_NCCL_ENABLE_MONITORING": "0", "TORCH_NCCL_DUMP_ON_TIMEOUT": "0"}
    ours = [k for k in disarm if k not in os.environ]
    for k in ours:
        os.environ[k] = disarm[k]
    try:
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=max(1800.0, 20.0 * timeout_s)))
    finally:
        for k in ours:
            os.environ.pop(k, None)
    dist.barrier(group=group, device_ids=[device.index])
    t = torch.tensor([float(dist.get_rank())], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    torch.cuda.synchronize(device)
    if int(t.item()) != dist.get_world_size() - 1:
        raise RuntimeError(f"RCCL MAX all-reduce returned {t.item()} for world {dist.get_world_size()}")
    return group


def rendezvous(world: int, rank: int, local_rank: int, *, prefer: str = "nccl", device=None, timeout_s: float = 90.0,
               allow_oversubscribe: bool = False, prove_nccl: Callable = _prove_nccl) -> Rendezvous:
    """Form the world (see the module docstring).  `device`: the torch.device this rank drives, or None on a CPU-only dry run.
    Reads MASTER_ADDR / MASTER_PORT from the environment (default address 127.0.0.1: the container's hostname may not resolve)."""
    if world <= 1:
        return Rendezvous(world=1, rank=0, backend="none", ranks_seen=[{"rank": 0, "local_rank": local_rank, "pid": os.getpid(), **_device_identity(device)}])
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(30.0, 2 * timeout_s)))
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "host": socket.gethostname(), **_device_identity(device)}
    seen: List[Optional[dict]] = [None] * world
    dist.all_gather_object(seen, me)
    seen = sorted(seen, key=lambda r: r["rank"])
    keys = [(r["device"], r["uuid"]) for r in seen]
    have_devices = all(r["device"] is not None for r in seen)
    oversub = have_devices and len(set(keys)) < world
    if oversub and not allow_oversubscribe:
        dist.destroy_process_group()
        raise RuntimeError(f"rendezvous: {world} ranks but only {len(set(keys))} distinct device(s): {seen} "
                           "(one rank per GPU; SPECTRE_BENCH_OVERSUBSCRIBE=1 allows a dry run on fewer)")
    rdv = Rendezvous(world=world, rank=rank, backend="gloo", ranks_seen=seen, oversubscribed=bool(oversub), _device=device)
    if prefer != "nccl":
        return rdv
    # every rank evaluates the same census, so the decision to try RCCL at all is already unanimous
    if not have_devices:
        rdv.fallback_reason = "no HIP device on at least one rank"
        return rdv
    if oversub:
        rdv.fallback_reason = "several ranks share one device (RCCL refuses duplicate GPUs)"
        return rdv
    result: dict = {}

    def attempt():
        try:
            result["group"] = prove_nccl(device, timeout_s)
        except BaseException as e:                   # noqa: BLE001 - whatever RCCL throws, the run goes on over gloo
            result["error"] = f"{type(e).__name__}: {e}"

    th = threading.Thread(target=attempt, daemon=True)
    th.start()
    th.join(timeout_s + 5.0)
    if th.is_alive():
        result.setdefault("error", f"RCCL group not proven within {timeout_s:.0f} s")
    ok_local = "group" in result and "error" not in result
    verdicts: List[Optional[dict]] = [None] * world
    dist.all_gather_object(verdicts, {"rank": rank, "ok": ok_local, "error": result.get("error")})
    if all(v["ok"] for v in verdicts):
        rdv.backend, rdv._nccl_group = "nccl", result["group"]
    else:
        bad = [v for v in verdicts if not v["ok"]]
        rdv.fallback_reason = "RCCL failed on rank(s) " + ", ".join(f"{v['rank']} ({str(v['error'])[:160]})" for v in bad)
    return rdv
