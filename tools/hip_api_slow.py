"""Developer aid: the slow HIP API calls of the last mapping-loop turn.
    rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d OUT -o t -- python tools/time_mapinc.py
    python tools/hip_api_slow.py OUT/t_hip_api_trace.csv [min_us]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
# the last turn: from the last hipMemcpyAsync that follows a pause of >= 1 ms (the host generating the next scan) to teardown
starts = [i for i in range(1, len(rows)) if rows[i][0] - rows[i - 1][1] > 1_000_000]
last = rows[starts[-1]:] if starts else rows
last = [r for r in last if not r[2].startswith(("hipHostFree", "hipFree", "hipStreamDestroy", "hipModuleUnload"))]
t0 = last[0][0]
prev_end = t0
for s, e, f in last:
    d = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    if d >= thr or gap >= thr:
        print("%9.1f us  gap %6.1f  dur %7.1f  %s" % ((s - t0) / 1e3, gap, d, f))
    prev_end = max(prev_end, e)
