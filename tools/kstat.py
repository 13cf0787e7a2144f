"""Developer aid: average durations of selected kernels from a rocprofv3 --stats kernel_stats.csv.
    python tools/kstat.py OUT/x_kernel_stats.csv vox_add tombstone ..."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+(<[\w, ]+>)?)", r["Name"])
    n = m.group(1) if m else r["Name"][:28]
    if len(sys.argv) < 3 or any(k in n for k in sys.argv[2:]):
        print("%-28s calls %4s avg %8.1f us" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3))
