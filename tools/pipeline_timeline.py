"""Developer aid: the last turns of tools/time_pipeline.py as a kernel / copy timeline with the queue of every activity.
    cd /tmp; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -o pipe -- python tools/time_pipeline.py
    python tools/pipeline_timeline.py OUT/pipe_kernel_trace.csv OUT/pipe_memory_copy_trace.csv [turn_us]"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("malio::", "").replace("void ", ""), r.get("Queue_Id", "?")))
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?").replace("MEMORY_COPY_", ""), "dma"))
ev.sort()
# the last occurrence of k_far_nearest<1> marks the start of a turn (map_incremental's first kernel)
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_far_nearest")]
a, b = starts[-3], starts[-2]
t0 = ev[a][0]
queues = sorted({e[3] for e in ev[a:b]})
print("turn (map_incremental k -> scan_set k+1 -> update k+1): %.1f us, queues %s" % ((ev[b][0] - t0) / 1e3, queues))
busy_end = t0
idle = 0.0
for s, e, n, q in ev[a:b]:
    gap = (s - busy_end) / 1e3
    if gap > 0: idle += gap
    print("%8.1f  %-6s %6.1f us  %s%s" % ((s - t0) / 1e3, str(q)[-4:], (e - s) / 1e3, n[:60], ("   <- %.1f us nothing running" % gap) if gap > 1.0 else ""))
    busy_end = max(busy_end, e)
print("nothing running for %.1f us of the turn" % idle)
