"""Per-kernel mean of every counter in rocprofv3 --pmc CSV output (counter_collection.csv files)."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(d + "/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[-40:] + " g" + r["Grid_Size"]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[-1])
    for k, cs in acc.items():
        if not any(x in k for x in ("search", "rows", "reuse", "final", "k_pass")): continue
        print("  %-52s" % k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
