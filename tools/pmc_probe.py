"""Round 5: which hardware counter separates the flat copy from a tile-shaped persistent kernel?

    python tools/pmc_probe.py --out gpurun_out/r05_pmc --tag window_lab -- tools/window_lab 2

Runs `rocprofv3 --list-avail` once (kept as <out>/counters_avail.txt), picks every counter of the wish list below that this
rocprofv3 / gfx950 exposes (plus anything whose name smells of address translation or the Infinity Cache), and runs the command under
SEPARATE `--pmc` passes of a few counters each (counters only: no trace domains, as the pool's gpurun demands).  The result is one
table per tag: rows = kernel names, columns = per-dispatch means of every counter that could be collected."""
import argparse
import collections
import csv
import glob
import os
import re
import subprocess
import sys

WISH = [
    # address translation (per-CU UTCL1 in the vector cache; anything deeper if exposed)
    ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_PERMISSION_MISS_sum"],
    ["TCP_UTCL1_STALL_INFLIGHT_MAX_sum", "TCP_UTCL1_STALL_LRU_INFLIGHT_sum", "TCP_UTCL1_STALL_MULTI_MISS_sum", "TCP_UTCL1_LFIFO_FULL_sum"],
    ["TCP_UTCL1_STALL_LFIFO_NOT_RES_sum", "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_NC_READ_REQ_sum"],
    ["TCP_TA_TCP_STATE_READ_sum", "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
    # L2 <-> fabric: how many requests, how many go to DRAM, how long they take, who stalls
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_32B_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_64B_sum"],
    ["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_RDREQ_IO_CREDIT_STALL_sum", "TCC_EA0_WRREQ_STALL_sum"],
    ["TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum"],
    ["TCC_TAG_STALL_sum", "TCC_BUBBLE_sum", "TCC_BUSY_sum", "TCC_CYCLE_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_NORMAL_WRITEBACK_sum"],
    ["TCC_READ_sum", "TCC_WRITE_sum", "TCC_NORMAL_EVICT_sum", "TCC_ALL_TC_OP_WB_WRITEBACK_sum"],
    ["TCC_EA0_RD_UNCACHED_32B_sum", "TCC_EA0_WR_UNCACHED_32B_sum", "TCC_EA0_ATOMIC_sum", "TCC_EA0_ATOMIC_LEVEL_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT", "FETCH_SIZE", "WRITE_SIZE"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INST_LEVEL_VMEM"],
    ["TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum", "TCP_UTCL1_THRASHING_STALL_sum", "TCP_UTCL1_SERIALIZATION_STALL_sum", "GRBM_UTCL2_BUSY"],
    ["TCC_EA0_RDREQ_DRAM_32B_sum", "TCC_EA0_WRREQ_WRITE_DRAM_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "TCC_IB_STALL_sum"],
]
SMELL = re.compile(r"UTCL2|TLB|MALL|VML2|ATC|XNACK|_EFC_|INFINITY", re.I)


def list_avail(out):
    path = os.path.join(out, "counters_avail.txt")
    if not os.path.exists(path):
        for flag in ("--list-avail", "-L", "--list-counters"):
            r = subprocess.run(["rocprofv3", flag], capture_output=True, text=True)
            txt = r.stdout + r.stderr
            if "TCC_" in txt or "SQ_" in txt:
                open(path, "w").write(txt)
                break
        else:
            open(path, "w").write("(no counter list obtained)\n" + txt)
    txt = open(path).read()
    names = set(re.findall(r"\b([A-Z][A-Za-z0-9]*_[A-Za-z0-9_]+)\b", txt))
    return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kernel-filter", default="")
    ap.add_argument("--max-passes", type=int, default=40)
    ap.add_argument("--passes", default="", help="comma-separated indices into the pass list (wish list order, then the extra counters)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    cmd = [os.path.abspath(c) if os.path.exists(c) else c for c in cmd]      # the passes run from /tmp
    a.out = os.path.abspath(a.out)
    os.makedirs(a.out, exist_ok=True)
    names = list_avail(a.out)
    passes = []
    for grp in WISH:
        have = [c for c in grp if (c in names or c.replace("_sum", "") in names or not names)]
        if have:
            passes.append(have)
    extra = sorted(n for n in names if SMELL.search(n) and not any(n in p for p in passes))
    for i in range(0, len(extra), 4):
        passes.append(extra[i:i + 4])
    if a.passes:
        passes = [passes[int(i)] for i in a.passes.split(",") if int(i) < len(passes)]
    passes = passes[:a.max_passes]
    missing = [c for grp in WISH for c in grp if names and c not in names and c.replace("_sum", "") not in names]
    log = open(os.path.join(a.out, f"{a.tag}_passes.log"), "w")
    print(f"counters on the wish list that this rocprofv3 does not expose: {missing}", file=log)
    print(f"address-translation / Infinity-Cache counters found by name: {extra}", file=log)
    table = collections.defaultdict(dict)
    env = dict(os.environ, TMPDIR="/tmp")
    for i, grp in enumerate(passes):
        d = os.path.join(a.out, f"{a.tag}_pass{i:02d}")
        r = subprocess.run(["rocprofv3", "--output-format", "csv", "--pmc", *grp, "-d", d, "-o", "pmc", "--", *cmd],
                           capture_output=True, text=True, env=env, cwd="/tmp")
        ok = False
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0]
                if a.kernel_filter and a.kernel_filter not in k:
                    continue
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                ok = True
        for k, cs in agg.items():
            for c, v in cs.items():
                table[k][c] = (sum(v) / len(v), len(v))
        print(f"pass {i}: {grp} -> rc {r.returncode}, {'collected' if ok else 'NOTHING collected'}", file=log)
        if not ok:
            print((r.stdout + r.stderr)[-1500:], file=log)
        log.flush()
        subprocess.run(["rm", "-rf", d])
        write_table(a, table)                       # after EVERY pass: a time-out must not lose what was collected
    print(open(os.path.join(a.out, f"{a.tag}_table.txt")).read()[-6000:])


def write_table(a, table):
    cols = sorted({c for k in table for c in table[k]})
    with open(os.path.join(a.out, f"{a.tag}_table.txt"), "w") as f:
        for k in sorted(table):
            print(f"== {k}   (dispatches per pass: {max(n for _, n in table[k].values())})", file=f)
            for c in cols:
                if c in table[k]:
                    print(f"   {c:48s} {table[k][c][0]:18.1f}", file=f)


if __name__ == "__main__":
    main()
