#!/usr/bin/env python3
"""bench.py — spectral-mix tokens/s at (B=256, N=4096, D=768) per GPU, batch-sharded over N GPUs.

    python bench.py [--gpus N --steps K --warmup W] [--io f32|bf16] [--shape B,N,D] [--groups G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (fused rfft -> gate -> irfft, spectre.py:506,:542-553) over one
synthetic batch already resident in HBM.  Weak scaling: every rank owns B=256 batch elements (the 8-GPU
configuration is BASELINE.json's B=2048 batch shard); there is no collective in the data path, only the
barrier + MAX-over-ranks timing reduction the contract asks for.

Prints ONE JSON line on rank 0 with
  value        whole-job tokens/s = B_total * N * K / (last rank's finish - first rank's start) of the K timed steps: every rank
               stamps the node's monotonic clock behind the opening barrier + synchronize and behind its closing synchronize (in front
               of the closing barrier), so the job time is MAX over ranks of the end minus MIN over ranks of the start — the
               max-over-ranks wall time without the latency of the closing barrier itself (`timing` in the line says so;
               `wall_incl_closing_barrier_s` is the older definition)
  roofline     dominant kernel vs the HBM roofline: algorithmic bytes per launch / average launch duration,
               the duration measured live with HIP events on the launch stream over the timed region
  cpu_baseline the oracle's torch.fft restatement of the same statements timed on this box's host cores
               on a bounded sample (rank 0, single-GPU runs only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copies: roofline.*_copy_GBps of the line this prints


def algorithmic_bytes(B, N_in, n_fft, D, G, es_in, es_out, mem=False):
    """one read of V, one write of out, the gate once (+ memory_fft once)."""
    F = n_fft // 2 + 1
    n_out = min(N_in, n_fft)
    return B * N_in * D * es_in + B * n_out * D * es_out + B * G * F * 8 + (F * D * 8 if mem else 0)


def cpu_baseline(N, D, G, seconds_budget=20.0):
    """torch.fft on the host cores: the reference's own statements (oracle.spectral_mix_torch), bounded sample."""
    import torch
    from oracle.spectral_mix_oracle import spectral_mix_torch
    cores = os.cpu_count() or 1
    try:
        torch.set_num_threads(cores)
    except Exception:
        pass
    Bc = 32
    g = torch.Generator().manual_seed(0)
    V = torch.randn(Bc, N, D, generator=g)
    F = N // 2 + 1
    gate = (torch.complex(torch.randn(Bc, G, F, generator=g), torch.randn(Bc, G, F, generator=g)) * 0.3)
    spectral_mix_torch(V, gate, None, N)                     # warm-up (MKL plan, allocator)
    reps, t0 = 0, time.perf_counter()
    while True:
        spectral_mix_torch(V, gate, None, N)
        reps += 1
        el = time.perf_counter() - t0
        if reps >= 3 and (el > seconds_budget * 0.5 or reps >= 20):
            break
        if el > seconds_budget:
            break
    return {"value": Bc * N * reps / el, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.spectral_mix_torch (torch.fft restatement of spectre.py:506,:542-553), fp32, "
                      f"B={Bc} N={N} D={D} G={G}, {reps} reps in {el:.1f} s"}


def launch_command(n_gpus: int, argv: list) -> list:
    """`python -m torch.distributed.run` command that runs this file as n_gpus ranks on one node."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


PMC_CHILD_LAUNCHES = 8


def pmc_child(a):
    """`bench.py --pmc-child`: the headline launch PMC_CHILD_LAUNCHES times on tensors of the headline's shape and nothing else — the
    target of the two rocprofv3 --pmc passes of live_traffic().  (Counters serialise the kernels and cost seconds per pass, so they are
    never collected around the timed region itself.)"""
    import torch
    from fft_amd import spectral_mix
    from fft_amd import _native
    _native.load()
    B, N, D = (int(x) for x in a.shape.split(","))
    dev = torch.device("cuda:0")
    dt = torch.float32 if a.io == "f32" else torch.bfloat16
    F = N // 2 + 1
    torch.manual_seed(0)
    V = torch.randn(B, N, D, device=dev).to(dt)
    gate = torch.randn(B, a.groups, F, dtype=torch.complex64, device=dev) * 0.3
    gate = gate * (torch.rand(B, a.groups, F, device=dev) >= 0.18)
    out = torch.empty_like(V)
    for _ in range(PMC_CHILD_LAUNCHES):
        spectral_mix(V, gate, None, N, out=out)
    torch.cuda.synchronize()


def parse_pmc_csv(root, counter, want="spectre_mix_"):
    """Per-dispatch values of `counter` for the kernels whose name contains `want`, from rocprofv3's *counter_collection.csv."""
    import csv
    import glob
    vals, names = [], set()
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if want in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                    vals.append(float(r["Counter_Value"]))
                    names.add(r["Kernel_Name"].split("(")[0][:96])
    return vals, sorted(names)


def live_traffic(a, timeout_s=150.0):
    """HBM bytes per launch of the headline kernel, measured in THIS bench invocation (the line used to replay
    profiles/pmc_latest.json): two separate `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass,
    MI355X_MICROARCH.md) over `bench.py --pmc-child`, after everything that is timed has been timed.  gfx950 correction as the guide
    prescribes: FETCH_SIZE counts a 128-byte request as 64 bytes -> doubled; WRITE_SIZE as is.  Returns (record | None, reason)."""
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("SPECTRE_BENCH_PMC", "1") == "0":
        return None, "SPECTRE_BENCH_PMC=0"
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under rocprofv3 (no nested collection)"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    got, names, t0 = {}, [], time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="spectre_pmc_", dir="/tmp") as td:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--output-format", "csv", "--pmc", c, "-d", os.path.join(td, c), "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--io", a.io, "--shape", a.shape, "--groups", str(a.groups)]
            try:
                p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            except OSError as e:
                return None, f"rocprofv3 could not be started: {e}"
            try:
                _, err = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)                  # our own session: rocprofv3 and the child it started
                except OSError:
                    pass
                p.wait()
                return None, f"the {c} pass did not finish within {timeout_s:.0f} s"
            if p.returncode != 0:
                return None, f"the {c} pass exited with {p.returncode}: {err.decode(errors='replace')[-200:]}"
            vals, nm = parse_pmc_csv(os.path.join(td, c), c)
            if not vals:
                return None, f"the {c} pass recorded no dispatch of the headline kernel"
            got[c] = vals
            names = nm
    fetch = sum(got["FETCH_SIZE"]) / len(got["FETCH_SIZE"])
    write = sum(got["WRITE_SIZE"]) / len(got["WRITE_SIZE"])
    return {"hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024, "FETCH_SIZE_kb": fetch, "WRITE_SIZE_kb": write,
            "launches": [len(got["FETCH_SIZE"]), len(got["WRITE_SIZE"])], "kernels": names,
            "seconds": time.perf_counter() - t0}, "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-placement-probe", action="store_true", help="skip the informational best-of-4-allocations variant")
    ap.add_argument("--prewarm", type=int, default=60,
                    help="untimed launches BEFORE the --warmup steps: the chip needs ~25 back-to-back launches (40 ms) to leave its idle "
                         "power state (tools/ramp_time.py, profiles/r03_ramp_time.log); reported as `prewarm_steps`")
    ap.add_argument("--io", choices=["f32", "bf16"], default="f32",
                    help="storage dtype of V and out (arithmetic is always fp32; the reference is fp32-only)")
    ap.add_argument("--shape", default="256,4096,768", help="per-GPU B,N,D")
    ap.add_argument("--groups", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--print-launch", action="store_true", help="print the multi-process launch command for --gpus N and exit")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic live (two rocprofv3 --pmc child passes, ~1 min); "
                                                          "the line then replays profiles/pmc_latest.json and says so")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.pmc_child:
        pmc_child(a)
        return

    # --gpus N without a launcher around us: become the launcher (one rank per GPU; rendezvous on 127.0.0.1: gloo control plane first,
    # then an RCCL group that has to prove itself — fft_amd/rendezvous.py)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = launch_command(a.gpus, [x for x in sys.argv[1:] if x != "--print-launch"])
        if a.print_launch:
            print(" ".join(cmd))
            return
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus and not os.environ.get("SPECTRE_BENCH_OVERSUBSCRIBE"):
            sys.exit(f"bench.py: --gpus {a.gpus} requested but only {have} HIP device(s) visible")
        import subprocess
        sys.exit(subprocess.call(cmd))

    import torch
    from fft_amd import describe, spectral_mix
    from fft_amd import _native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if local >= torch.cuda.device_count() and not os.environ.get("SPECTRE_BENCH_OVERSUBSCRIBE"):
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local}, {torch.cuda.device_count()} visible)")
    oversub = bool(os.environ.get("SPECTRE_BENCH_OVERSUBSCRIBE"))
    local = local % max(1, torch.cuda.device_count())           # (several ranks on one GPU only in oversubscribed dry runs)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    # Rank rendezvous (fft_amd/rendezvous.py): gloo control plane first, then an RCCL group proven with a barrier + MAX all-reduce inside a
    # time box; any failure on any rank -> every rank keeps gloo for the barrier / MAX (no tensor data crosses ranks in this workload).
    # SPECTRE_BENCH_BACKEND=gloo skips the RCCL attempt.  The line says which transport ran (`rendezvous`) and who drove what (`ranks_seen`).
    from fft_amd.rendezvous import rendezvous
    rdv = rendezvous(world, rank, local, prefer=os.environ.get("SPECTRE_BENCH_BACKEND", "nccl"), device=dev,
                     timeout_s=float(os.environ.get("SPECTRE_BENCH_RDV_TIMEOUT", "90")), allow_oversubscribe=oversub)
    _native.load()                                            # fail loudly if the HIP library is missing

    B, N, D = (int(x) for x in a.shape.split(","))
    G = a.groups
    dt = torch.float32 if a.io == "f32" else torch.bfloat16
    F = N // 2 + 1
    # synthetic inputs, SURVEY.md §8(d): V ~ N(0,1); gate ~ 0.3 CN(0,1) with ~18 % exact zeros (modReLU)
    torch.manual_seed(rank)
    V = torch.randn(B, N, D, device=dev).to(dt)
    gate = torch.randn(B, G, F, dtype=torch.complex64, device=dev) * 0.3
    gate = gate * (torch.rand(B, G, F, device=dev) >= 0.18)
    out = torch.empty_like(V)
    kernel = describe(V, gate, None, N, out=out)

    def step():
        spectral_mix(V, gate, None, N, out=out)

    def barrier():
        rdv.barrier()
        torch.cuda.synchronize()

    # The contract's protocol WITHOUT the power-state prewarm, from the idle state the process starts in (the protocol of rounds 1-2, ADVICE
    # r03): W warm-up + K timed steps.  Informational (`cold_start`): round-over-round comparisons must use like for like.  Every world
    # size runs it (rank 0 reports its own), and its launches COUNT as prewarm launches, so that the number of launches in front of the
    # timed region is the same for every N and every round (`launches_before_timed_region`).
    cold = None
    launched = 0
    if a.prewarm >= a.warmup + a.steps:
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(a.steps):
            step()
        c1.record()
        torch.cuda.synchronize()
        cold = c0.elapsed_time(c1) / a.steps
        launched = a.warmup + a.steps
    for _ in range(max(0, a.prewarm - launched)):             # power-state ramp (untimed; the contract's warmup steps follow)
        step()
    launched = max(launched, a.prewarm)
    for _ in range(a.warmup):
        step()
    launched += a.warmup
    clock = rdv.node_clock                                    # CLOCK_MONOTONIC: one clock for every process of the node
    barrier()                                                 # the contract's opening barrier + synchronize
    rdv.common_start()                                        # ... and a common start (fft_amd/rendezvous.py; tests/test_multigpu_gloo.py)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = clock()
    ev0.record()                                              # same stream the kernels are launched on
    for _ in range(a.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    t_end = clock()
    rdv.barrier()                                             # the contract's closing barrier (the synchronize is in front of the stamp)
    wall_old = clock() - t_start
    kern_ms_own = ev0.elapsed_time(ev1) / a.steps             # average launch duration over the timed region, this rank
    kernel_timed = describe(V, gate, None, N, out=out)        # (the tile order of the persistent kernels is measured per shape class during the prewarm)
    wall_old, kern_ms = rdv.max_over_ranks([wall_old, kern_ms_own])
    window = rdv.job_window(t_start, t_end)
    wall = window["wall_s"]                                   # whole job: first rank's start -> last rank's finish
    per_rank = rdv.gather_over_ranks({"rank": rank, "kernel_ms": kern_ms_own, "wall_s": t_end - t_start,
                                      "tile_order": kernel_timed.rsplit("order=", 1)[-1] if "order=" in kernel_timed else None,
                                      "start_after_first_us": window["start_after_first_us"],
                                      "end_before_last_us": window["end_before_last_us"]})

    # other storage dtypes of the same workload, same run (informational; `value` above is the --io default).
    # BASELINE.json configs[2] words the headline shape as "bf16 in / fp32 compute": both readings are reported.
    variants = {}
    ceilings = {}
    VARIANT_WARMUP = 45        # untimed launches per informational variant: the tensor conversions in between let the power state drop (--prewarm)
    if world == 1:
        from fft_amd import copy_probe, time_kernel
        # what a PURE COPY of the same bytes reaches on this device, in this process (C ABI spectre_probe_copy, fft_amd/csrc/copy_probe.hip):
        #   dense_copy   = 256-KiB contiguous chunks, 16 bytes per lane
        #   pattern_copy = 128-byte row segments of 4096 consecutive rows at the tensor's row stride: what the product kernel presents to
        #                  the memory system once the two workgroups of a pair have merged their 64-byte halves in the L2 (PMC: all reads
        #                  leave the L2 as 128-byte requests, 1.06 write-backs per line) — the ceiling of its access pattern
        #   half_line_copy = the same with free-running 64-byte (fp32) / 32-byte (bf16) segments per workgroup, neighbours on the other pieces
        #                  of the lines: what the pattern costs when NOTHING keeps the neighbours together (the product beats it)
        if V.is_contiguous() and (B * N * D * V.element_size()) % (256 * 1024) == 0 and N % 4096 == 0:
            seg = 16 * V.element_size()
            byt = 2.0 * B * N * D * V.element_size()
            for name, sg, md in (("dense_copy", 0, "copy"), ("pattern_copy", 128, "copy"), ("half_line_copy", seg, "copy"),
                                 ("pattern_load_only", 128, "load"), ("pattern_store_only", 128, "store"), ("dense_load_only", 0, "load"),
                                 ("half_line_load_only", seg, "load"), ("half_line_store_only", seg, "store")):
                best = None
                for per_cu in (1, 2, 4):                                  # persistent workgroups per CU: report the best of three
                    ms = copy_probe(V, out, sg, mode=md, wgs_per_cu=per_cu, warmup=3, iters=max(5, a.steps // 2))
                    best = ms if best is None or ms < best else best
                nb = byt if md == "copy" else byt / 2
                ceilings[name + "_GBps"] = nb / best / 1e6
                ceilings[name + "_ms"] = best
            # Round 4: the forms the hardware guide's 6.29 TB/s comes from — NON-persistent, one 256-thread workgroup
            # per 4 KiB, 16 bytes per lane — plain and with non-temporal accesses, and the runtime's own device-to-device copy.  None of
            # them has the product's access pattern (64-byte row segments of 4096 rows at the row stride): `best_copy_GBps` is what the
            # memory system gives ANY kernel on this (V, out) pair, `pattern_copy_GBps` what it gives this tile shape.
            for name, sg in (("flat_copy", -1), ("flat_nt_copy", -2), ("hipMemcpyDtoD", -3)):
                ms = copy_probe(V, out, sg, mode="copy", warmup=3, iters=max(5, a.steps // 2))
                ceilings[name + "_GBps"] = byt / ms / 1e6
                ceilings[name + "_ms"] = ms
            # What bounds the product (round 4): a workgroup owns 16 channels = HALF a 128-byte line per row, and the L2 takes a half-line
            # STORE at about two thirds of the rate of a full-line one (store-only probes: 3.4-3.6 TB/s in 64-byte segments against 5.3-5.5 in
            # 128-byte segments; loads 5.3-5.5 against 5.9-6.1) although every byte still reaches HBM exactly once (PMC: all write requests
            # leave the L2 as full 64-byte bursts).  The product's own requests, issued alone, take half_line_load_only_ms +
            # half_line_store_only_ms; `frac_of_half_line_requests` = that sum / the product's time (1 = the two directions do not overlap
            # in the request path at all, > 1 = they do).
            ceilings["half_line_requests_ms"] = ceilings["half_line_load_only_ms"] + ceilings["half_line_store_only_ms"]
            ceilings["best_copy_GBps"] = max(ceilings[k] for k in ("dense_copy_GBps", "pattern_copy_GBps", "flat_copy_GBps", "flat_nt_copy_GBps", "hipMemcpyDtoD_GBps"))
            for _ in range(10):
                step()                                                    # (the probes overwrote `out`; leave a valid result behind)
        for name, tin, tout in (("bf16_in_bf16_out", torch.bfloat16, torch.bfloat16), ("bf16_in_f32_out", torch.bfloat16, torch.float32),
                                ("f32_in_f32_out", torch.float32, torch.float32)):
            if (tin == dt and tout == dt):
                continue
            Vv = V.to(tin)
            ov = torch.empty(B, N, D, dtype=tout, device=dev)
            ms = time_kernel(Vv, gate, None, N, out=ov, warmup=VARIANT_WARMUP, iters=max(3, a.steps // 2))
            byt = algorithmic_bytes(B, N, N, D, G, Vv.element_size(), ov.element_size())
            variants[name] = {"tokens_per_s": B * N / (ms * 1e-3), "kernel_ms": ms, "achieved_GBps": byt / ms / 1e6,
                              "roofline_frac": byt / ms / 1e6 / HBM_PEAK_GBS, "kernel": describe(Vv, gate, None, N, out=ov)}
            # the variant's OWN pattern ceiling: bf16 rows are 32-byte segments, four neighbouring workgroups per
            # 128-byte line (spectre_probe_copy seg 32, gang of 4), measured on the variant's own tensors
            if tin == tout and tin != dt and Vv.is_contiguous() and (B * N * D * Vv.element_size()) % (256 * 1024) == 0 and N % 4096 == 0:
                cb = 2.0 * B * N * D * Vv.element_size()
                pm = min(copy_probe(Vv, ov, 16 * Vv.element_size(), mode="copy", wgs_per_cu=w, warmup=3, iters=max(5, a.steps // 2)) for w in (1, 2, 4, 8))
                fm = copy_probe(Vv, ov, -1, mode="copy", warmup=3, iters=max(5, a.steps // 2))
                variants[name].update({"pattern_copy_GBps": cb / pm / 1e6, "pattern_copy_ms": pm, "frac_of_pattern_copy": pm / ms * (byt / cb),
                                       "flat_copy_GBps": cb / fm / 1e6})
            # mixed storage (bf16 rows in, fp32 rows out: BASELINE configs[2] read literally) has no copy of its own shape; its ceiling is its
            # two request streams issued alone, each in the kernel's own segment width on the variant's own tensors: 32-byte loads of the
            # bf16 rows (gangs of four per line) + 64-byte stores of the fp32 rows (pairs) — VERDICT r04 item 3
            if tin != tout and Vv.is_contiguous() and N % 4096 == 0 and (B * N * D * Vv.element_size()) % (256 * 1024) == 0:
                ldm = min(copy_probe(Vv, Vv, 16 * Vv.element_size(), mode="load", wgs_per_cu=w, warmup=3, iters=max(5, a.steps // 2)) for w in (1, 2, 4))
                stm = min(copy_probe(ov, ov, 16 * ov.element_size(), mode="store", wgs_per_cu=w, warmup=3, iters=max(5, a.steps // 2)) for w in (1, 2, 4))
                variants[name].update({"pattern_load_only_ms": ldm, "pattern_store_only_ms": stm, "pattern_requests_ms": ldm + stm,
                                       "frac_of_pattern_requests": (ldm + stm) / ms,
                                       "pattern_note": f"{16 * Vv.element_size()}-byte load-only pass over V + {16 * ov.element_size()}-byte store-only pass over out, "
                                                       "issued one after the other; > 1 = the kernel overlaps its two directions"})
            del Vv, ov
        # Where the driver places an allocation is worth +-5 % to this kernel and +-10 % to a copy (LABNOTES.md section 5, round 3, item 7;
        # tools/placement_time.py): allocations fall into two classes, one of which takes stores ~19 % faster.  The headline above uses the
        # tensors as torch allocated them (no shopping).  Informational: the same launch on the fastest of 4 candidate allocations each for
        # V and out, chosen by the load-only / store-only copy probes — what an application that probes its long-lived buffers would see.
        if V.is_contiguous() and (B * N * D * V.element_size()) % (256 * 1024) == 0 and not a.no_placement_probe:
            try:
                cands_in = [V] + [V.clone() for _ in range(3)]
                cands_out = [out] + [torch.empty_like(out) for _ in range(3)]
                ld = [min(copy_probe(c, out, 0, mode="load", wgs_per_cu=w, warmup=5, iters=10) for w in (2, 4)) for c in cands_in]
                st = [min(copy_probe(V, c, 0, mode="store", wgs_per_cu=w, warmup=5, iters=10) for w in (2, 4)) for c in cands_out]
                bi, bo = min(range(4), key=lambda i: ld[i]), min(range(4), key=lambda i: st[i])
                ms = time_kernel(cands_in[bi], gate, None, N, out=cands_out[bo], warmup=VARIANT_WARMUP, iters=max(3, a.steps // 2))
                byt = algorithmic_bytes(B, N, N, D, G, V.element_size(), out.element_size())
                variants["best_of_4_allocations"] = {"tokens_per_s": B * N / (ms * 1e-3), "kernel_ms": ms, "achieved_GBps": byt / ms / 1e6,
                                                     "roofline_frac": byt / ms / 1e6 / HBM_PEAK_GBS, "dense_load_only_ms": ld, "dense_store_only_ms": st,
                                                     "chosen": [bi, bo], "note": "index 0 = the tensors of the headline"}
                del cands_in, cands_out
            except torch.OutOfMemoryError:
                pass
        # the other single-GPU configurations of BASELINE.json / SURVEY.md section 8(d), informational (C1 and C4)
        for name, (Bc, Nc, Dc) in (("C1_f32_256x1024x768", (256, 1024, 768)), ("C4_f32_256x3000x768", (256, 3000, 768))):
            Vc = torch.randn(Bc, Nc, Dc, device=dev)
            gc = torch.randn(Bc, G, Nc // 2 + 1, dtype=torch.complex64, device=dev) * 0.3
            oc = torch.empty_like(Vc)
            ms = time_kernel(Vc, gc, None, Nc, out=oc, warmup=VARIANT_WARMUP, iters=max(3, a.steps // 2))
            byt = algorithmic_bytes(Bc, Nc, Nc, Dc, G, 4, 4)
            variants[name] = {"tokens_per_s": Bc * Nc / (ms * 1e-3), "kernel_ms": ms, "achieved_GBps": byt / ms / 1e6,
                              "roofline_frac": byt / ms / 1e6 / HBM_PEAK_GBS, "kernel": describe(Vc, gc, None, Nc, out=oc)}
            # this configuration's own pattern ceiling: a pure copy in the kernel's tile shape on its own tensors — whole 128-byte lines
            # of all Nc rows for the whole-line tiles of n_fft <= 1024 (kernel_regtile_wide.h), 64-byte halves in pairs otherwise
            if (Bc * Nc * Dc * 4) % (256 * 1024) == 0:
                sgc = 128 if "wide" in variants[name]["kernel"] else 64
                if True:                                                  # (tiles that are not whole 128-KiB chunks — C4: 64 B x 3000 rows — run the probe's masked form)
                    pm = min(copy_probe(Vc, oc, sgc, tile_rows=Nc, mode="copy", wgs_per_cu=w, warmup=3, iters=max(5, a.steps // 2)) for w in (1, 2, 4))
                    variants[name].update({"pattern_copy_ms": pm, "pattern_copy_GBps": 2.0 * Bc * Nc * Dc * 4 / pm / 1e6,
                                           "frac_of_pattern_copy": pm / ms * (byt / (2.0 * Bc * Nc * Dc * 4)), "pattern_segment_bytes": sgc})
            del Vc, gc, oc
        # backward of the same op (row N1), informational: dV = the forward kernels with conj(gate); dgate = gate-gradient kernel
        from fft_amd import spectral_mix_backward
        dout = torch.randn(B, N, D, device=dev).to(dt)
        for name, kw in (("backward_dV", dict(need_dv=True, need_dgate=False)), ("backward_dgate", dict(need_dv=False, need_dgate=True))):
            for _ in range(VARIANT_WARMUP):                                # untimed: first call builds the plan, the rest ramp the power state
                spectral_mix_backward(V, gate, dout, N, **kw)              # and let the (dOut, dV) pair's tile order settle (40 launches)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, a.steps // 2)
            e0.record()
            for _ in range(reps):
                spectral_mix_backward(V, gate, dout, N, **kw)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            byt = 2 * B * N * D * V.element_size()     # dV: read dOut, write dV; dgate: read V and dOut
            variants[name] = {"tokens_per_s": B * N / (ms * 1e-3), "kernel_ms": ms, "achieved_GBps": byt / ms / 1e6,
                              "roofline_frac": byt / ms / 1e6 / HBM_PEAK_GBS}
        # the gate gradient reads V and dOut in 32-byte row segments (8 fp32 channels per tile): what a LOAD-ONLY pass with that pattern
        # reaches here (C ABI spectre_probe_copy, 32-byte segments of 4096 rows), next to the kernel
        if V.is_contiguous() and V.dtype == torch.float32 and (B * N * D * 4) % (256 * 1024) == 0 and N % 4096 == 0:
            ld = min(copy_probe(V, out, 32, mode="load", wgs_per_cu=w, warmup=5, iters=10) for w in (1, 2, 4))
            g32 = B * N * D * 4 / ld / 1e6
            variants["backward_dgate"].update({"load_only_32B_segments_GBps": g32, "frac_of_32B_load_only": variants["backward_dgate"]["achieved_GBps"] / g32})
        del dout
        # the launch between the heads' concatenation and out_proj in the reference's DEFAULT layer (WaveletRefinement, spectre.py:819-887; beyond
        # SURVEY section 8), informational: in place on the mix's output, the reference's default rate 0.1 and every element on
        try:
            from fft_amd import wavelet_refine
            if N & (N - 1) == 0 and out.dtype in (torch.float32, torch.bfloat16):
                wgate = torch.rand(B, D, device=dev)
                wgen = torch.Generator(device=dev).manual_seed(7)
                for wname, rate in (("wavelet_refine_on_rate_0.1", 0.1), ("wavelet_refine_all_on", 1.0)):
                    wmask = torch.rand(B, device=dev, generator=wgen) < rate
                    n_on = int(wmask.sum())
                    for _ in range(5):
                        wavelet_refine(out, wgate, wmask, inplace=True)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        wavelet_refine(out, wgate, wmask, inplace=True)
                    e1.record(); torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 10
                    byt = 2 * n_on * N * D * out.element_size()          # rows of the switched-on elements in and out
                    variants[wname] = {"kernel_ms": ms, "elements_on": n_on, "achieved_GBps": byt / ms / 1e6 if ms > 0 else 0.0,
                                       "roofline_frac": byt / ms / 1e6 / HBM_PEAK_GBS if ms > 0 else 0.0}
                for _ in range(3):
                    step()                                                # (the refinement overwrote `out`; leave a valid result behind)
        except Exception as exc:                                          # informational: never take the line down
            variants["wavelet_refine_on_rate_0.1"] = {"error": repr(exc)}

    if rank == 0:
        es = V.element_size()
        alg = algorithmic_bytes(B, N, N, D, G, es, es)
        achieved = alg / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source, traffic_live = None, None, None
        pmc_why = "--no-pmc" if a.no_pmc else ("N > 1" if world != 1 else None)
        if pmc_why is None:
            traffic_live, pmc_why = live_traffic(a)               # behind every timed launch of this process
        if traffic_live is not None:
            traffic = traffic_live["hbm_bytes_per_launch"]
            traffic_source = (f"measured in this run: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --pmc-child` "
                              f"({traffic_live['launches'][0]} + {traffic_live['launches'][1]} launches of {', '.join(traffic_live['kernels'])} on tensors of this shape, "
                              f"{traffic_live['seconds']:.0f} s, after the timed region); FETCH_SIZE doubled (gfx950 counts a 128-byte request as 64 bytes), per-launch average")
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")  # written by tools/collect_pmc.py on the GPU box
        if traffic is None and os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("io") == a.io and rec.get("shape") == [B, N, D]:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/pmc_latest.json: separate rocprofv3 --pmc passes of this command "
                                      "(tools/collect_pmc.py), kernel " + str(rec.get("kernel", "?")) + f"; not collected in this run ({pmc_why})")
            except Exception:
                traffic = None
        res = {
            "metric": "spectral-mix tokens/sec at (B=256, N=4096, D=768) per GPU, batch-sharded",
            "value": world * B * N * a.steps / wall,
            "unit": "tokens/s",
            "n_gpus": world,
            "tokens_per_s_per_gpu": B * N * a.steps / wall,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"spectral-mix forward (rfft->gate->irfft), per-GPU (B={B}, N={N}, D={D}), G={G}, "
                                   f"n_fft={N}, {a.io} in/out, fp32 arithmetic; global batch {world * B}",
                       "io_dtype": a.io, "global_batch": world * B, "seq_len": N, "d_model": D,
                       "parallelism": f"batch-shard x{world} (no collective)", "kernel": kernel_timed, "kernel_before_first_launch": kernel},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_over_algorithmic": (traffic / alg if traffic else None),
                         "algorithmic_bytes_per_launch": alg, "kernel_ms": kern_ms},
            "prewarm_steps": max(0, a.prewarm),
            "launches_before_timed_region": launched,
            "timing": "value = tokens / (MAX over ranks of the finish stamp - MIN over ranks of the start stamp); stamps = the node's "
                      "CLOCK_MONOTONIC behind barrier + synchronize (start) and behind the closing synchronize, in front of the closing "
                      "barrier (finish); roofline.kernel_ms = MAX over ranks of the HIP-event average of the K launches",
            "wall_s": wall,
            "wall_incl_closing_barrier_s": wall_old,
            # rounds 1-4 defined `value` over the wall time INCLUDING the closing barrier: reported next to `value` so that round-over-round and
            # N-GPU comparisons with those rounds stay like for like
            "value_incl_closing_barrier": world * B * N * a.steps / wall_old,
            "per_rank": sorted(per_rank, key=lambda r: r["rank"]),
            **rdv.describe(),
        }
        if cold is not None:
            # the contract's literal protocol as a first-class number next to roofline.frac: like for like across rounds
            res["cold_start_frac"] = alg / (cold * 1e-3) / 1e9 / HBM_PEAK_GBS
            res["cold_start_ms"] = cold
            res["cold_start"] = {"kernel_ms": cold, "tokens_per_s": B * N / (cold * 1e-3), "roofline_frac": alg / (cold * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "protocol": f"{a.warmup} warm-up + {a.steps} timed launches from the idle state (rounds 1-2 protocol); they count as "
                                             f"the first {a.warmup + a.steps} of the {a.prewarm} prewarm launches"}
        if ceilings:
            res["roofline"].update(ceilings)
            res["roofline"]["frac_of_pattern_copy"] = achieved / ceilings["pattern_copy_GBps"]
            res["roofline"]["frac_of_dense_copy"] = achieved / ceilings["dense_copy_GBps"]
            res["roofline"]["frac_of_best_copy"] = achieved / ceilings["best_copy_GBps"]
            res["roofline"]["frac_of_half_line_requests"] = ceilings["half_line_requests_ms"] / kern_ms
            res["roofline"]["ceilings_note"] = ("pure copies of this launch's V -> out bytes measured in this process through the C ABI "
                                                "(spectre_probe_copy): dense = contiguous 16 B per lane; pattern = 128-byte row segments of 4096 rows "
                                                "at the row stride (the product's 64-byte halves merged per pair of workgroups); half_line = free-running "
                                                "64-byte (fp32) segments; best of 1 / 2 / 4 persistent workgroups per CU.  flat / flat_nt = the NON-persistent float4 copy "
                                                "(one 256-thread workgroup per 4 KiB; the form MI355X_MICROARCH.md quotes at 6.29 TB/s), plain / non-temporal; "
                                                "hipMemcpyDtoD = hipMemcpyAsync; best_copy = the fastest of all forms")
        if variants:
            res["variants"] = variants
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(N, D, G)
        print(json.dumps(res), flush=True)
    hung = bool(rdv.fallback_reason and "not proven within" in rdv.fallback_reason)
    if not hung:
        rdv.close()
    sys.stdout.flush()
    if hung:
        os._exit(0)                                            # an RCCL call that never returned still holds a thread: do not wait for it


if __name__ == "__main__":
    main()
