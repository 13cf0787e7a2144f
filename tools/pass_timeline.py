"""Per-pass timeline from a rocprofv3 --kernel-trace CSV for any pass shape: passes are cut at the kernels named in
FIRST (k_search, k_reuse, k_pass); prints, per distinct kernel sequence, the median duration of each kernel, the gap
before it and the period from pass start to the next pass start.
    python tools/pass_timeline.py OUT/x_kernel_trace.csv"""
import csv, sys, collections, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
FIRST = ("k_search<", "k_reuse", "k_pass")
KEEP = ("k_search", "k_reuse", "k_pass", "k_rows_reduce", "k_final_reduce", "k_search_tail")
def short(n):
    n = n.replace("void ", "").replace("malio::", "").split("(")[0]
    return n if any(n.startswith(k) for k in KEEP) else None
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = [s for s in seq if s[0]]
passes, cur = [], []
for s in seq:
    if any(s[0].startswith(f) for f in FIRST) and not s[0].startswith("k_search_tail"):
        if cur: passes.append(cur)
        cur = [s]
    elif cur:
        cur.append(s)
if cur: passes.append(cur)
groups = collections.defaultdict(list)
for i in range(len(passes) - 1):
    p, nxt = passes[i], passes[i + 1]
    kind = tuple(k[0] for k in p)
    if kind[0].startswith("k_pass") or kind[0].startswith("k_search<true"):  # one kernel name, two kinds of pass
        kind = kind + ("[search]" if p[0][2] - p[0][1] > 18000 else "[reuse]",)
    groups[kind].append((p, nxt[0][1]))
for shape, items in groups.items():
    if len(items) < 20: continue
    items = items[len(items) // 4:]
    print("pass shape %s  (%d passes)" % (" -> ".join(shape), len(items)))
    for j, n in enumerate(k for k in shape if not k.startswith("[")):
        d = st.median(p[j][2] - p[j][1] for p, _ in items) / 1e3
        g = st.median((p[j][1] - p[j - 1][2]) for p, _ in items) / 1e3 if j else 0.0
        print("   %-26s dur %7.2f us   gap before %6.2f us" % (n, d, g))
    span = st.median(p[-1][2] - p[0][1] for p, _ in items) / 1e3
    turn = st.median(nx - p[-1][2] for p, nx in items) / 1e3
    print("   first start -> last end %7.2f us;  last end -> next pass' first start (host turnaround) %6.2f us" % (span, turn))
