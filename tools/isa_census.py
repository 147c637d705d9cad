"""Static instruction census of the shipped kernels of one translation unit, from hipcc's device listing (the whole kernel body: prologue +
tile loop + epilogue; the tile loop is > 95 % of it for the persistent kernels).  Classes: VALU arithmetic (v_add / v_sub / v_mul / v_fma /
v_mac / v_pk_*), VALU moves and conversions, cross-lane (v_permlane*, DPP), LDS (ds_*), vector memory (buffer_* / global_*), scalar ALU, scalar
memory, waits, barriers.

    python tools/isa_census.py regtile_n4096p.hip [name filter]  > profiles/r05_isa_census.txt
"""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import isa_lint
from fft_amd.build import CXXFLAGS, CSRC


def cls(op, text):
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vector memory"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("v_"):
        if "dpp" in text or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")):
            return "VALU cross-lane"
        if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_mad_f32", "v_fmamk_f32", "v_fmaak_f32", "v_pk_")):
            return "VALU fp32 arithmetic"
        if op.startswith(("v_mov", "v_cvt", "v_accvgpr", "v_perm", "v_and", "v_or", "v_lshl", "v_lshr", "v_bfe", "v_cndmask")):
            return "VALU move / convert / bit"
        return "VALU other (integer, compare, address)"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_barrier":
        return "s_barrier"
    if op.startswith(("s_load", "s_buffer_load", "s_atomic", "s_memtime", "s_memrealtime", "s_dcache")):
        return "scalar memory"
    return "scalar ALU / control"


def main():
    unit = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    flags = [f for f in CXXFLAGS if f not in ("-fPIC",)]
    listing = isa_lint.compile_asm(os.path.join(CSRC, unit), flags)
    kernels = isa_lint.parse_kernels(listing)
    names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.split("\n")
    print(f"# static instruction census, {unit} (hipcc device listing, flags {' '.join(flags)})")
    order = ["VALU fp32 arithmetic", "VALU move / convert / bit", "VALU cross-lane", "VALU other (integer, compare, address)", "LDS", "vector memory",
             "scalar ALU / control", "scalar memory", "s_waitcnt", "s_barrier"]
    for mangled, name in zip(kernels, names):
        if filt and filt not in name:
            continue
        c = collections.Counter()
        ops = collections.Counter()
        for b in kernels[mangled]:
            for ins in b["ins"]:
                k = cls(ins.op, ins.text)
                c[k] += 1
                if k == "VALU fp32 arithmetic":
                    ops[ins.op.split("_e")[0]] += 1
        tot = sum(c.values())
        print(f"\n== {name.replace('sfft::', '')}\n   {tot} instructions")
        for k in order:
            if c[k]:
                print(f"   {k:40s} {c[k]:6d}")
        print("   fp32 arithmetic by opcode: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(8)))
    os.unlink(listing)


if __name__ == "__main__":
    main()
