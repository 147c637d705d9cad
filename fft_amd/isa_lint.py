"""Build-time ISA lint for the hand-counted waits of the pipelined 4096 kernel (fft_amd/csrc/kernel_regtile64p.h).

The kernel reads its LDS-DMA landing slots behind `s_waitcnt vmcnt(N)` with N = the number of VMEM instructions the wave issues
between the last `buffer_load ... lds` of a burst and that wait (completion is in order, so "at most N outstanding" means the DMA has
landed).  N is a constant in the source (p64_younger / p64_younger_first); this script recounts it in what hipcc actually emitted:

  * compiles the translation unit to gfx950 assembly (hipcc -S --offload-device-only),
  * builds the control-flow graph of every kernel whose name matches --kernel,
  * finds the guards = `s_waitcnt vmcnt(N)` that come from inline asm (bracketed by ;;#ASMSTART / ;;#ASMEND),
  * walks BACKWARDS from each guard over every path until it meets an LDS-DMA instruction, counting VMEM instructions on the way,
  * fails (exit 1) if on any path fewer than N VMEM instructions separate the guard from the DMA (the wait would be too loose: a wave
    could read a slot that has not landed), and reports paths with MORE than N (the wait is stricter than necessary: performance only).

A guard taken in the steady state is checked against the DMA instructions inside the tile loop, the guard of the first iteration
against the prologue's.  The source says which is which with a comment inside the asm statement (`s_waitcnt vmcnt(N) ; lint: steady` /
`; lint: first` — the comment survives into the -S listing); an untagged pair is told apart by N (the smaller count is the prologue's).

    python -m fft_amd.isa_lint fft_amd/csrc/regtile_n4096p.hip [--kernel regtile64p] [--asm out.s] [--flags "..."]

A second check (--check lds) covers the untracked LDS reads of kernel_regtile_mixedp.h: its exchanges read the image with inline-asm
`ds_read_b32` whose results hipcc's s_waitcnt insertion does not know about; they are consumed behind rt_lds_barrier() (s_waitcnt
lgkmcnt(0) ; s_barrier) and pinned there.  The script follows every such read forward through the listing and fails if any instruction
mentions its destination register before an `s_waitcnt` with lgkmcnt(0) has been passed.  The same kernels write the image with
`ds_write_addtid_b32` (base in M0, set inside each asm statement): --check addtid (part of --check lds) verifies that every one follows
its own `s_mov_b32 m0` and that no compiler-generated instruction of the kernel touches M0.

    python -m fft_amd.isa_lint fft_amd/csrc/regtile_mixedp.hip --kernel mixedp --check lds
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
DEFAULT_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"]

VMEM = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
KERNEL = re.compile(r"^(_Z\w+):\s*;\s*@")
BRANCH = re.compile(r"^s_(c?branch\w*)\s+(\.LBB\d+_\d+)")
WAIT_VM = re.compile(r"^s_waitcnt\b.*vmcnt\((\d+)\)")


class Ins:
    __slots__ = ("op", "text", "inline", "is_vmem", "is_dma", "guard", "tag")

    def __init__(self, text, inline, comment=""):
        self.text = text
        mt = re.search(r"lint:\s*(first|steady)", comment)
        self.tag = mt.group(1) if mt else None
        self.op = text.split()[0]
        self.inline = inline
        self.is_vmem = bool(VMEM.match(self.op))
        self.is_dma = self.is_vmem and text.rstrip().endswith(" lds")
        m = WAIT_VM.match(text)
        self.guard = int(m.group(1)) if (m and inline) else None
        if self.guard == 0 and re.search(r"lint:\s*drain", comment):
            self.guard = None                      # an explicit full drain (nothing in flight afterwards) guards no landing slot: always safe


def compile_asm(src, flags):
    fd, out = tempfile.mkstemp(suffix=".s")
    os.close(fd)
    cmd = [HIPCC, *flags, "--offload-device-only", "-S", src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed:\n{r.stderr}")
    return out


def parse_kernels(path):
    """-> {kernel name: [blocks]}, block = dict(label, ins, succ (labels / 'FALL'))"""
    kernels, cur, blocks, blk, inline = {}, None, None, None, False
    for raw in open(path):
        line = raw.strip()
        mk = KERNEL.match(line)
        if mk:
            cur = mk.group(1)
            blocks = kernels[cur] = []
            blk = {"label": "ENTRY", "ins": [], "order": 0}
            blocks.append(blk)
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if line.startswith(";;#ASMSTART"):
            inline = True
            continue
        if line.startswith(";;#ASMEND"):
            inline = False
            continue
        ml = LABEL.match(line)
        if ml:
            blk = {"label": ml.group(1), "ins": [], "order": len(blocks)}
            blocks.append(blk)
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.startswith("//"):
            continue
        code, _, comment = line.partition(";")
        code = code.strip()
        if not code:
            continue
        ins = Ins(code, inline, comment)
        blk["ins"].append(ins)
        if ins.op.startswith("s_cbranch") or ins.op in ("s_branch", "s_endpgm"):
            # a branch ends the basic block; the fall-through part gets an anonymous block
            blk = {"label": f"{blk['label']}+{len(blocks)}", "ins": [], "order": len(blocks)}
            blocks.append(blk)
    return kernels


def build_cfg(blocks):
    by_label = {b["label"]: i for i, b in enumerate(blocks)}
    preds = [[] for _ in blocks]
    for i, b in enumerate(blocks):
        last = b["ins"][-1] if b["ins"] else None
        targets = []
        if last is not None and last.op == "s_endpgm":
            pass
        elif last is not None and last.op == "s_branch":
            targets.append(by_label[BRANCH.match(last.text).group(2)])
        else:
            if last is not None and last.op.startswith("s_cbranch"):
                targets.append(by_label[BRANCH.match(last.text).group(2)])
            if i + 1 < len(blocks):
                targets.append(i + 1)
        for t in targets:
            preds[t].append(i)
    return preds


def walk_back(blocks, preds, bi, ii):
    """All (count, dma block order) over paths that end right before instruction ii of block bi and start behind an LDS-DMA instruction."""
    results = []
    best = {}          # (block, entry index) -> smallest count seen (a larger count through the same point cannot lower the minimum)
    worst = {}
    stack = [(bi, ii, 0)]
    while stack:
        b, i, cnt = stack.pop()
        ins = blocks[b]["ins"]
        j = i - 1
        hit = False
        while j >= 0:
            x = ins[j]
            if x.is_dma:
                results.append((cnt, blocks[b]["order"]))
                hit = True
                break
            if x.is_vmem:
                cnt += 1
            j -= 1
        if hit:
            continue
        for pb in preds[b]:
            key = (pb, len(blocks[pb]["ins"]))
            lo, hi = best.get(key), worst.get(key)
            if lo is not None and lo <= cnt and hi >= cnt:
                continue                     # both extremes through this point are already covered
            best[key] = cnt if lo is None else min(lo, cnt)
            worst[key] = cnt if hi is None else max(hi, cnt)
            if cnt > 4096:
                raise RuntimeError("runaway path (a loop without LDS-DMA between the guard and itself)")
            stack.append((pb, len(blocks[pb]["ins"]), cnt))
    return results


def lint_kernel(name, blocks, verbose=True):
    preds = build_cfg(blocks)
    guards = [(bi, ii, x.guard, x.tag) for bi, b in enumerate(blocks) for ii, x in enumerate(b["ins"]) if x.guard is not None]
    dma_orders = sorted({b["order"] for b in blocks if any(x.is_dma for x in b["ins"])})
    if not guards:
        return [f"{name}: no inline-asm s_waitcnt vmcnt(N) found (the guard is gone?)"], []
    if len(dma_orders) < 2:
        return [f"{name}: expected LDS-DMA in the prologue and in the tile loop, found blocks {dma_orders}"], []
    first_dma = dma_orders[0]            # prologue (text order)
    errors, notes = [], []
    values = sorted({g[2] for g in guards})
    if len(values) == 1 and len(guards) > 1 and not all(g[3] for g in guards):
        errors.append(f"{name}: {len(guards)} untagged guards with the same count vmcnt({values[0]}): tag them `; lint: first` / `; lint: steady`")
    for bi, ii, n, tag in guards:
        paths = walk_back(blocks, preds, bi, ii)
        # steady-state guard <-> DMA of the tile loop; first-tile guard <-> DMA of the prologue
        steady = (tag == "steady") if tag else ((n == values[-1]) if len(values) > 1 else True)
        rel = [c for c, o in paths if (o != first_dma) == steady]
        if not rel:
            errors.append(f"{name}: guard vmcnt({n}) is not reachable from {'the loop' if steady else 'the prologue'} LDS-DMA")
            continue
        lo, hi = min(rel), max(rel)
        kind = "steady state" if steady else "first tile"
        if lo < n:
            errors.append(f"{name}: guard vmcnt({n}) [{kind}] TOO LOOSE: a path has only {lo} VMEM instructions behind the last LDS-DMA")
        notes.append(f"{name}: guard vmcnt({n}) [{kind}]: VMEM instructions behind the last LDS-DMA on any path: min {lo}, max {hi}"
                     + ("" if hi == n else f"  (paths with more than {n}: stricter than needed, not wrong)"))
    return errors, notes


VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT_LGKM0 = re.compile(r"^s_waitcnt\b(?!.*lgkmcnt\([1-9])(.*lgkmcnt\(0\)|\s+0\s*$)")


def _vregs(text):
    regs = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def lint_lds_reads(name, blocks):
    """Every inline-asm ds_read_b32: no instruction may mention its destination VGPR before an s_waitcnt lgkmcnt(0) (text order: the reads
    sit in a lane-predicated block that falls through to the barrier)."""
    flat = [x for b in blocks for x in b["ins"]]
    errors, reads = [], 0
    for i, x in enumerate(flat):
        if not (x.inline and x.op == "ds_read_b32"):
            continue
        reads += 1
        dst = _vregs(x.text.split(",")[0])
        addr = _vregs(x.text.split(",")[1]) if "," in x.text else set()
        if dst & addr:
            errors.append(f"{name}: `{x.text}` overwrites its own address register (later reads of the burst use it)")
        for y in flat[i + 1:]:
            if WAIT_LGKM0.match(y.text):
                break
            if y.inline and y.op == "ds_read_b32":
                if dst & _vregs(y.text.split(",")[0]):
                    errors.append(f"{name}: two reads of one burst land in v{sorted(dst)[0]}")
                continue
            if dst & _vregs(y.text):
                errors.append(f"{name}: `{y.text}` touches v{sorted(dst)[0]} before the s_waitcnt lgkmcnt(0) behind `{x.text}`")
                break
        else:
            errors.append(f"{name}: no s_waitcnt lgkmcnt(0) behind `{x.text}`")
    notes = [f"{name}: {reads} untracked ds_read_b32, each consumed behind an s_waitcnt lgkmcnt(0)"]
    if reads == 0:
        errors.append(f"{name}: no inline-asm ds_read_b32 found (the exchange reads are gone?)")
    return errors, notes


def lint_addtid(name, blocks):
    """ds_write_addtid_b32 takes its base from M0, which the asm statement sets itself (hipcc treats M0 as reserved).  Each one must follow
    an s_mov_b32 m0 inside the same asm statement, and no compiler-generated instruction of the kernel may READ M0 (hipcc setting it — in front
    of an LDS-DMA request — is fine: the statements declare M0 clobbered; a compiler-generated read, s_movrel say, would depend on them)."""
    errors, n = [], 0
    for b in blocks:
        prev_inline_m0 = False
        for x in b["ins"]:
            mentions_m0 = bool(re.search(r"(?<![\w.])m0(?![\w.])", x.text))
            if x.op == "ds_write_addtid_b32":
                n += 1
                if not (x.inline and prev_inline_m0):
                    errors.append(f"{name}: `{x.text}` is not preceded by s_mov_b32 m0 inside its asm statement")
            if mentions_m0 and not x.inline:
                # hipcc may SET M0 (it does in front of every LDS-DMA request: the asm statements list M0 as clobbered, so it never assumes
                # a value survives them); anything else — a read, a read-modify-write — would depend on what an asm statement left there
                ops = [t.strip().rstrip(",") for t in x.text.split(None, 1)[1].split(",")] if " " in x.text else []
                sets_m0 = x.op in ("s_mov_b32", "s_add_i32", "s_add_u32", "s_lshl_b32", "s_or_b32", "s_and_b32") and ops and ops[0] == "m0" and "m0" not in ops[1:]
                if not sets_m0:
                    errors.append(f"{name}: compiler-generated `{x.text}` uses M0 in a kernel whose asm statements overwrite it")
            if x.inline and x.op == "s_mov_b32" and x.text.split()[1].rstrip(",") == "m0":
                prev_inline_m0 = True
            elif not (x.inline and x.op in ("s_nop", "ds_write_addtid_b32")):
                prev_inline_m0 = False
    if n == 0:
        errors.append(f"{name}: no ds_write_addtid_b32 found")
    return errors, [f"{name}: {n} ds_write_addtid_b32, each behind its own s_mov_b32 m0; no other use of M0"]


# ---- serialised global loads -------------------------------------------------------------------------------------------------------
# hipcc sometimes turns a run of independent loads (each unpacked or range-checked right behind it) into `load ; s_waitcnt vmcnt(0)`
# pairs: one memory request in flight per wave, a 2 x slowdown that no parity test can see (round 3: the bf16 loads of five mixed-radix
# lengths and of the bf16 gate gradient).  Rounds 3-5 guarded against it with a TIMING test on the GPU (bf16 rows must not take 1.4 x the
# fp32 rows' time); since round 6 the listing itself is checked at build time: the longest run of loads that are each followed by a full
# vmcnt(0) wait before the next load, per kernel.  KNOWN_SERIAL_RUNS = what the shipped listings contain today (secondary kernels: padded
# N = 2000, bf16 memory_fft mode at 640, the sweep of the mixed-radix ticket forms); a run of 8 or more anywhere else fails the build.
_LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load)_\w+\s")
_IS_LDS_DMA = re.compile(r"\blds\s*$")
_WAIT0 = re.compile(r"^\s*s_waitcnt\b.*vmcnt\(0\)")
_KERNEL = re.compile(r"^(_Z\w+):\s*;\s*@")
SERIAL_MIN_RUN = 8
SERIAL_EXEMPT = ("stockham",)                          # the general LDS fallback (kernel_stockham.h, the two-pass gate gradient): its global loads
                                                       # are twiddle / chirp look-ups inside unrolled radix loops, consumed where they are requested
KNOWN_SERIAL_RUNS = {                                  # mangled-name substring -> longest run accepted
    # the ticket protocol's own loads (first two tickets of a gang, the sweep: each depends on the one before; not on the data path)
    "spectre_mix_regtile64pILi3ELi3ELb0ELb0ELb0ELb1ELb1ELi1E": 10,
    "spectre_mix_regtile64pILi5ELi3ELb0ELb1ELb0ELb1ELb1ELi1E": 10,
    "spectre_mix_regtile64pILi5ELi3ELb0ELb1ELb1ELb1ELb1ELi1E": 13,
    "spectre_mix_regtile64pILi4ELi1ELb1ELb0ELb0ELb1ELb1ELi1E": 13,
    "spectre_mix_regtile_mixedpILi64ELi60ELi22ELb0ELi16ELi8ELb1E": 13,
    "spectre_mix_regtile_mixedpILi64ELi48ELi22ELb0ELi16ELi8ELb1E": 13,
    "spectre_mix_regtile_mixedpILi60ELi60ELi28ELb0ELi16ELi8ELb1E": 13,
    "spectre_mix_regtile_mixedpILi60ELi50ELi28ELb0ELi16ELi8ELb1E": 13,
    # secondary kernels (performance debt, not correctness): n_fft = 2000 (50 x 40: 100 data registers of the 128 a two-workgroups-per-CU
    # kernel may use) with padded fp32 rows, bf16 rows + memory_fft at 640.  Cause, from the listings: the
    # lane's row index is SPILLED, and every twiddle-base load is preceded by a scratch reload of it plus `s_waitcnt vmcnt(0)` — which also
    # waits for the previous base.  (Requesting all bases first and pinning them made no difference: the spill is the allocator's.  Applying the two twiddle factors in two
    # passes, each with its own bases, took it out of the fast and padded modes of 50 x 40 and into its mode 1 and into 40 x 30 bf16 +
    # memory_fft (58 loads): not kept.  regtile_n2000.hip keeps the NaN patch of the bf16 stores because WITHOUT it the same spill appears in
    # its bf16 -> bf16 fast mode: 1.375 -> 1.48 ms at (384, 2000, 768).)
    "spectre_mix_regtile_mixedILi50ELi40ELb0ELb0ELi3E": 26,
    "spectre_mix_regtile_mixedILi32ELi20ELb1ELb1ELi2E": 22,
}


def serial_load_runs(asm_path):
    """{kernel: longest run of `load ; s_waitcnt vmcnt(0)` pairs} for every kernel of a listing (LDS-DMA requests are not loads here)."""
    out, cur, run, best, pending = {}, None, 0, 0, False
    for line in open(asm_path):
        m = _KERNEL.match(line)
        if m:
            if cur:
                out[cur] = best
            cur, run, best, pending = m.group(1), 0, 0, False
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[cur] = best
            cur = None
            continue
        if _LOAD.match(line) and not _IS_LDS_DMA.search(line.split(";")[0]):
            if pending:                                # two loads without a full wait in between: the run is broken
                run = 0
            pending = True
        elif _WAIT0.match(line):
            if pending:
                run += 1
                best = max(best, run)
                pending = False
    if cur:
        out[cur] = best
    return out


def lint_serial_loads(asm_path, min_run=SERIAL_MIN_RUN, known=None):
    known = KNOWN_SERIAL_RUNS if known is None else known
    errors, notes = [], []
    for name, best in serial_load_runs(asm_path).items():
        if best < min_run or any(x in name for x in SERIAL_EXEMPT):
            continue
        allowed = max([v for k, v in known.items() if k in name] or [0])
        if best > allowed:
            errors.append(f"{name}: {best} global loads each behind its own s_waitcnt vmcnt(0) (one request in flight per wave)"
                          + (f"; the known run of this kernel is {allowed}" if allowed else ""))
        else:
            notes.append(f"{name}: known serialised run of {best} loads (accepted: {allowed})")
    return errors, notes


def lint_listing(asm_path, vmcnt_kernel=None, lds_kernel=None):
    """Every check that applies to one gfx950 listing (hipcc -S, or the .s that -save-temps leaves behind) -> (errors, notes).

    * every kernel that contains a `ds_write_addtid_b32` gets the M0 check (no name filter: all mixed-radix forward and gate-gradient
      kernels use those asm statements);
    * kernels whose name contains `vmcnt_kernel` get the hand-counted-guard check, those containing `lds_kernel` the untracked-read check;
    * every kernel: no run of SERIAL_MIN_RUN or more serialised global loads beyond KNOWN_SERIAL_RUNS."""
    errors, notes = lint_serial_loads(asm_path)
    for name, blocks in parse_kernels(asm_path).items():
        if any(x.op == "ds_write_addtid_b32" for b in blocks for x in b["ins"]):
            e, n = lint_addtid(name, blocks)
            errors += e
            notes += n
        if vmcnt_kernel and vmcnt_kernel in name:
            e, n = lint_kernel(name, blocks)
            errors += e
            notes += n
        if lds_kernel and lds_kernel in name:
            e, n = lint_lds_reads(name, blocks)
            errors += e
            notes += n
    return errors, notes


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help=".hip translation unit (or a .s file produced by hipcc -S)")
    ap.add_argument("--kernel", default="regtile64p", help="substring of the (mangled) kernel names to check")
    ap.add_argument("--flags", default=None, help="compiler flags (default: the library's)")
    ap.add_argument("--check", default="vmcnt", choices=["vmcnt", "lds", "addtid"],
                    help="vmcnt: hand-counted guards of LDS-DMA slots; lds: untracked ds_read_b32 (+ addtid); addtid: M0-based LDS writes")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    asm, tmp = args.source, None
    if not args.source.endswith(".s"):
        asm = tmp = compile_asm(args.source, args.flags.split() if args.flags else DEFAULT_FLAGS)
    try:
        kernels = {k: v for k, v in parse_kernels(asm).items() if args.kernel in k}
    finally:
        if tmp:
            os.unlink(tmp)
    if not kernels:
        print(f"isa_lint: no kernel matching '{args.kernel}'", file=sys.stderr)
        return 1
    bad = 0
    for name, blocks in kernels.items():
        if args.check == "vmcnt":
            errors, notes = lint_kernel(name, blocks)
        elif args.check == "addtid":
            errors, notes = lint_addtid(name, blocks)
        else:
            errors, notes = lint_lds_reads(name, blocks)
            e2, n2 = lint_addtid(name, blocks)
            errors, notes = errors + e2, notes + n2
        if not args.quiet:
            for n in notes:
                print("isa_lint:", n)
        for e in errors:
            print("isa_lint: ERROR", e, file=sys.stderr)
        bad += len(errors)
    if bad == 0 and not args.quiet:
        print(f"isa_lint: {len(kernels)} kernel(s) ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
