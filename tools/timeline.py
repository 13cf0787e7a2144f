"""Per-pass timeline from a rocprofv3 --kernel-trace CSV: mean duration of each kernel of the search pass and the
mean gap before it (end of the previous kernel of the same pass -> its start), plus pass-to-pass period."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["k_search", "k_rows_reduce", "k_final_reduce"]
def short(n):
    for s in names:
        if s in n:
            return s
    return None
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = [s for s in seq if s[0]]
passes, cur = [], []
for s in seq:
    if s[0] == names[0]:
        if len(cur) == len(names):
            passes.append(cur)
        cur = [s]
    elif cur:
        cur.append(s)
if len(cur) == len(names):
    passes.append(cur)
passes = passes[len(passes) // 4:]  # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list); period = []
for i, p in enumerate(passes):
    for j, (n, s, e) in enumerate(p):
        dur[n].append(e - s)
        if j: gap[n].append(s - p[j - 1][2])
    if i: period.append(p[0][1] - passes[i - 1][0][1]); gap[names[0]].append(p[0][1] - passes[i - 1][-1][2])
import statistics as st
print("passes analysed:", len(passes))
for n in names:
    print("%-16s dur %7.2f us   gap before %7.2f us" % (n, st.median(dur[n]) / 1e3, st.median(gap[n]) / 1e3 if gap[n] else 0))
print("pass period (median) %.2f us; sum of kernel durations %.2f us" % (st.median(period) / 1e3, sum(st.median(dur[n]) for n in names) / 1e3))
