"""Developer aid: one steady-state turn of the mapping loop (scan_set -> update -> map_incremental) as a kernel /
copy timeline. Usage (on the GPU box, from /tmp):
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -o loop -- python tools/time_mapinc.py
    python tools/loop_timeline.py OUT/loop_kernel_trace.csv [OUT/loop_memory_copy_trace.csv]
Prints every activity of the LAST turn (from its first k_scan* / copy after the previous turn's last kernel) with
start offset, duration and the idle gap before it, then per-name totals."""
import csv, sys, collections
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("malio::", "")))
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?") + ":" + r.get("Size", "?")))
ev.sort()
# turns are separated by the host-side scene generation: the largest idle gaps
gaps = sorted(((ev[i][0] - ev[i - 1][1], i) for i in range(1, len(ev))), reverse=True)
nturns = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cuts = sorted(i for _, i in gaps[:nturns - 1])
last = ev[cuts[-1]:]
t0 = last[0][0]
prev = t0
tot = collections.OrderedDict()
for s, e, n in last:
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n[:70]))
    prev = max(prev, e)
    d = tot.setdefault(n.split(":")[0] if n.startswith("copy") else n, [0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3
print("turn span %.1f us, busy %.1f us" % ((prev - t0) / 1e3, sum(v[1] for v in tot.values())))
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %-40s x%-4d %9.1f us" % (n[:40], c, d))
