"""Build libspectre_hip.so (gfx950) in-tree with hipcc.

    python -m fft_amd.build [--force]

The heavy register-resident kernels live in one translation unit per R so they compile in parallel.
The shared object lands in fft_amd/lib/ (git-ignored, but it travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libspectre_hip.so")

SOURCES = ["spectre_hip.hip", "copy_probe.hip", "wavelet.hip", "regtile_n4096.hip", "regtile_n4096p.hip", "regtile_n2048.hip", "regtile_n1024.hip", "regtile_n512.hip",
           "regtile_n256.hip", "regtile_wide.hip", "regtile_n3000.hip", "regtile_mixedp.hip", "regtile_n768.hip", "regtile_n1536.hip",
           "regtile_n3072.hip", "regtile_n1000.hip", "regtile_n2000.hip", "regtile_n1280.hip", "regtile_n2560.hip", "regtile_n3840.hip",
           "regtile_mixed_small.hip", "regtile_mixed_mid.hip", "regtile_mixed_mid2.hip", "regtile_n2400.hip", "regtile_n3600.hip", "regtile_n8192.hip", "regtile_n6144.hip", "regtile_n16384.hip", "regtile_n12288.hip"]
HEADERS = ["fft_regs.h", "fft_regs_mixed.h", "fft_tables_mixed.h", "kernel_regtile.h", "kernel_regtile64p.h", "kernel_tickets.h", "kernel_regtile_wide.h", "kernel_regtile_grad.h", "kernel_regtile_mixed.h", "kernel_regtile_mixedp.h", "kernel_regtile_mixed_grad.h", "kernel_regtile_long.h", "kernel_regtile_long_grad.h", "kernel_regtile_quad.h", "kernel_stockham.h", "kernel_gate.h", "kernel_gate_grad_twopass.h", "kernel_decode.h", os.path.join("..", "..", "include", "spectre_hip.h")]

# headers only ONE translation unit includes: a change rebuilds that unit, not all of them
OWN_HEADERS = {"wavelet.hip": ["kernel_wavelet.h"]}

# -fno-slp-vectorize: SLP packs the butterflies into v_pk_*_f32 (no faster than two scalar ops on gfx950)
# plus register-pair shuffles, which pushes the 64-point kernel past 256 VGPRs into scratch.
# -Wno-inline-asm: ds_write_addtid_b32 takes its base from M0, which the asm statements set and list as clobbered; clang warns that M0
# is a reserved register (it neither preserves it nor depends on it across asm statements — which is what those statements assume).
CXXFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall",
            "-Wno-unused-function", "-Wno-inline-asm"]


def _weight(src: str) -> int:
    """Rough compile cost of a translation unit: its largest transform length."""
    import re
    m = re.search(r"(\d+)", src)
    n = int(m.group(1)) if m else 0
    return {2048: 5000, 4096: 4500, 3000: 4000}.get(n, n)


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


# ISA lint (fft_amd/isa_lint.py): every translation unit is compiled with -save-temps, and the gfx950 listing that leaves behind is
# checked BEFORE the object is accepted — the build FAILS on a finding (a compiler that splits, merges or reorders one of the instructions
# the hand-written synchronisation counts on would otherwise turn it into a silent race):
#   * every kernel of every unit that contains a `ds_write_addtid_b32` (all mixed-radix forward and gate-gradient kernels): each one behind
#     its own `s_mov_b32 m0`, no compiler-generated use of M0 (-Wno-inline-asm hides clang's own warning about M0);
#   * the units below additionally: hand-counted `s_waitcnt vmcnt(N)` in front of the LDS-DMA landing slots / inline-asm ds_read_b32
#     consumed behind `s_waitcnt lgkmcnt(0)`.
LINTED = {"regtile_n4096p.hip": {"vmcnt_kernel": "regtile64p"},
          "regtile_mixedp.hip": {"lds_kernel": "mixedp", "vmcnt_kernel": "mixedp"}}


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.join(HERE, "isa_lint.py")]
    deps += [os.path.join(CSRC, h) for hs in OWN_HEADERS.values() for h in hs]
    return _newest(deps) > os.path.getmtime(LIB)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()

    hdr_time = _newest([os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.abspath(__file__), os.path.join(HERE, "isa_lint.py")])

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        own = [os.path.getmtime(os.path.join(CSRC, h)) for h in OWN_HEADERS.get(src, [])]
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_time, os.path.getmtime(os.path.join(CSRC, src)), *own):
            return obj                                # up to date (every translation unit includes most of the headers)
        # -save-temps=obj in a scratch directory of its own: the device listing (.s) is a by-product of the compile that the lint reads
        tmp = os.path.join(OBJDIR, "tmp_" + src.replace(".hip", ""))
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        tobj = os.path.join(tmp, os.path.basename(obj))
        cmd = [cc, *CXXFLAGS, "-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", tobj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            shutil.rmtree(tmp, ignore_errors=True)
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        listing = os.path.join(tmp, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        try:
            if not os.path.exists(listing):
                raise RuntimeError(f"ISA lint: hipcc left no device listing for {src} ({listing})")
            from . import isa_lint                     # part of the package: an installed / vendored copy lints its own builds
            errors, notes = isa_lint.lint_listing(listing, **LINTED.get(src, {}))
            if verbose:
                for n in notes:
                    print(f"isa_lint[{src}]:", n, flush=True)
            if src in LINTED and not notes:
                errors.append(f"{src}: the lint found none of the kernels it is meant to check ({LINTED[src]})")
            if errors:
                raise RuntimeError(f"ISA lint failed for {src}:\n" + "\n".join(errors))
            os.replace(tobj, obj)                      # accepted: only now does an object exist that the next build takes as up to date
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return obj

    # heaviest translation units first (64-point kernels), so the long poles do not start last
    order = sorted(SOURCES, key=lambda f: -_weight(f))
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        built = dict(zip(order, ex.map(compile_one, order)))
    objs = [built[f] for f in SOURCES]
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
