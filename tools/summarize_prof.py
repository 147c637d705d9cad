"""Condense rocprofv3 CSV output (kernel stats + PMC counters) into a short text summary."""
import csv, glob, os, sys, collections

root = sys.argv[1]
def find(pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))

print(f"# rocprofv3 summary for {root}")
for f in find("*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, root)})")
    for i, row in enumerate(csv.reader(open(f))):
        if i < 16:
            print(", ".join(c[:100] for c in row))
for f in find("*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"][:96]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"\n## kernel trace durations (us) ({os.path.relpath(f, root)})")
    for k, v in by.items():
        v2 = sorted(v)
        print(f"{k}: n={len(v)} min={v2[0]:.1f} median={v2[len(v2)//2]:.1f} max={v2[-1]:.1f} avg={sum(v)/len(v):.1f}")
    if rows:
        r = rows[-1]
        print("last dispatch: " + ", ".join(f"{k}={r[k]}" for k in r if k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")))
for f in find("*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"][:96]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n## counters ({os.path.relpath(f, root)})  [per-dispatch average]")
    for k, cs in agg.items():
        if "spectre" not in k and "tile" not in k:
            continue
        for c, v in cs.items():
            print(f"{k}: {c} = {sum(v)/len(v):.6g}  (n={len(v)})")
