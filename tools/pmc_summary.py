"""Sum the rocprofv3 --pmc counter CSVs under a directory per (kernel, counter):  python tools/pmc_summary.py dir [kernel-substring]"""
import csv, glob, os, sys, collections, json
root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            if pat and pat not in k:
                continue
            k = k.split("(")[0][-60:]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k][row["Counter_Name"]] += 1
out = {k: {c: {"sum": v, "per_dispatch": v / calls[k][c], "dispatches": calls[k][c]} for c, v in sorted(cs.items())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
