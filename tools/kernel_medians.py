"""developer aid: median / min / max duration per kernel name of a rocprofv3 --kernel-trace csv (a --stats average hides a
one-off long call, e.g. the list sort after a map build among the per-batch ones)
   python tools/kernel_medians.py <kernel_trace.csv> [name substring ...]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
by = {}
for r in rows:
    n = r["Kernel_Name"]
    if pats and not any(p in n for p in pats):
        continue
    by.setdefault(n.split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(by.items()):
    print("%-44s n %4d  median %7.1f  min %7.1f  max %8.1f us" % (n[:44], len(v), statistics.median(v), min(v), max(v)))
