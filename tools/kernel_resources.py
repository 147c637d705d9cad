"""Code-object metadata of every kernel in fft_amd/lib/obj/*.o: VGPRs, AGPRs, SGPRs, spills, scratch, static LDS, waves/SIMD allowed
by the registers.  The numbers DESIGN.md quotes come from here (rocprofv3's VGPR_Count column reports an allocation granule,
not the kernel's own count).

    python tools/kernel_resources.py [pattern]            # e.g. regtile64p, n4096, gate_grad
"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else ""


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


rows = []
for obj in sorted(glob.glob(os.path.join(ROOT, "fft_amd", "lib", "obj", "*.o"))):
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "x.fatbin"), os.path.join(td, "x.co")
        if subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fb}", obj], capture_output=True).returncode:
            continue
        if subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           f"--input={fb}", f"--output={co}"], capture_output=True).returncode:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        f = {m.group(1): m.group(2).strip() for m in re.finditer(r"\.(\w+):\s+(.+)", blk)}
        if "name" not in f:
            continue
        rows.append((os.path.basename(obj), f["name"], int(f.get("vgpr_count", 0)), int(f.get("agpr_count", 0)), int(f.get("sgpr_count", 0)),
                     int(f.get("vgpr_spill_count", 0)), int(f.get("sgpr_spill_count", 0)), int(f.get("private_segment_fixed_size", 0)),
                     int(f.get("group_segment_fixed_size", 0)), int(f.get("max_flat_workgroup_size", 0))))
dm = demangle([r[1] for r in rows])
print(f"{'object':24s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'waves/SIMD':>10s}  kernel")
for (obj, name, v, ag, sg, vs, ss, scr, lds, wg) in rows:
    d = dm.get(name, name)
    if pat and pat not in d and pat not in obj:
        continue
    alloc = (v + ag + 7) // 8 * 8
    waves = min(8, 512 // max(alloc, 8))
    d = re.sub(r"sfft::", "", d)[:110]
    print(f"{obj:24s} {v:4d} {ag:4d} {sg:4d} {vs:6d} {ss:6d} {scr:7d} {lds:6d} {waves:10d}  {d}")
