"""Turn the rocprofv3 PMC passes of tools/profile.sh into profiles/pmc_latest.json (read by bench.py's
`roofline.traffic`).  HBM bytes per launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024: on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is taken as is."""
import csv, glob, json, os, sys

root, io, shape = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
want = sys.argv[4] if len(sys.argv) > 4 else "spectre_mix"          # substring of the kernel name the counters are taken from
vals, kname = {}, None
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            kname = r["Kernel_Name"].split("(")[0][:80]
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
rec = {"io": io, "shape": shape, "kernel": kname, "FETCH_SIZE_kb": fetch, "WRITE_SIZE_kb": write,
       "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
       "note": "FETCH_SIZE doubled (gfx950 counts 128-B requests at 64 B), separate --pmc passes, per-dispatch average"}
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_latest.json")
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
