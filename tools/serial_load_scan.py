"""Scan every translation unit of the library for SERIALISED global loads: runs of `load ; s_waitcnt vmcnt(0)` pairs, i.e. one memory
request in flight per wave.  hipcc did that to the bf16 loads of the RF = 48 / 64 mixed-radix kernels in round 3 (a 2x slowdown that no
test can see).  Compiles each .hip with -S (in parallel) and reports, per kernel, the longest run of loads that are each followed by a
full vmcnt(0) wait before the next load.  Exit code 1 if a run of 8 or more is found.

    python tools/serial_load_scan.py [--min-run 8] [file.hip ...]
"""
import argparse, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import build as B     # noqa: E402  (SOURCES, CXXFLAGS, hipcc())

LOAD = re.compile(r"^\s*(global_load|buffer_load|flat_load)_\w+\s")
IS_LDS_DMA = re.compile(r"\blds\s*$")
WAIT0 = re.compile(r"^\s*s_waitcnt\b.*vmcnt\(0\)")
ANYWAIT = re.compile(r"^\s*s_waitcnt\b.*vmcnt\((\d+)\)")
KERNEL = re.compile(r"^(_Z\w+):\s*;\s*@")


def scan(asm_path):
    out, cur, run, best, pending = {}, None, 0, 0, False
    for line in open(asm_path):
        m = KERNEL.match(line)
        if m:
            if cur: out[cur] = best
            cur, run, best, pending = m.group(1), 0, 0, False
            continue
        if cur is None: continue
        if line.startswith(".Lfunc_end"):
            out[cur] = best; cur = None; continue
        if LOAD.match(line) and not IS_LDS_DMA.search(line.split(";")[0]):
            if pending:              # two loads without a full wait in between: the run is broken
                run = 0
            pending = True
        elif WAIT0.match(line):
            if pending:
                run += 1; best = max(best, run); pending = False
        elif ANYWAIT.match(line):
            pass                     # a partial wait neither completes nor breaks a pair
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--min-run", type=int, default=8)
    args = ap.parse_args()
    srcs = args.files or [os.path.join(B.CSRC, s) for s in B.SOURCES]
    cc = B.hipcc()
    tmp = tempfile.mkdtemp(prefix="serial_scan_")
    def one(src):
        out = os.path.join(tmp, os.path.basename(src) + ".s")
        flags = [f for f in B.CXXFLAGS if f != "-fPIC"]
        r = subprocess.run([cc, *flags, "--offload-device-only", "-S", src, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            return src, None, r.stderr[-400:]
        return src, scan(out), ""
    bad = 0
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        for src, res, err in ex.map(one, srcs):
            if res is None:
                print(f"{os.path.basename(src)}: compile failed: {err}"); bad += 1; continue
            worst = sorted(((v, k) for k, v in res.items() if v >= args.min_run), reverse=True)
            print(f"{os.path.basename(src)}: {len(res)} kernels, longest serialised run {max(res.values(), default=0)}")
            for v, k in worst:
                d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
                print(f"    {v:3d} loads each behind its own vmcnt(0): {d[:150]}"); bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
