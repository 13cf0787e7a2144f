"""HBM traffic per launch of the dominant kernel from one tools/profile_round.sh output directory:
  python tools/pmc_traffic.py gpurun_out/<tag> <tag> [kernel]
reads <tag>_pmc_summary.txt (FETCH_SIZE / WRITE_SIZE means, KB) and <tag>_bench_kernel_stats.csv (rocprofv3 average
duration) and writes <tag>_pmc_traffic.json - the file bench.py cites as `roofline.traffic` / `rocprof_kernel_ms`.
Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 64 B per TCC_EA0_RDREQ although wide
(16 B/lane) coalesced reads move 128 B per request -> doubled (an upper bound: the scattered 16 B gathers are not wide)."""
import ast, csv, json, os, re, sys
d, tag = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "k_pass"
rnd = sys.argv[4] if len(sys.argv) > 4 else "round6"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge; ge.load_package()
from malio_amd import capi
build_id = capi.lib().malio_build_id().decode()  # the binary the counters were taken from (this run's library)
vals = {}
for line in open(os.path.join(d, tag + "_pmc_summary.txt")):
    m = re.match(r"\s+malio::(\w+)(<[\w, ]+>)? g\d+\s+(\{.*\})", line)
    # (<false, false>: state in the kernel arguments, FULL search - the bench's step; <false, true> keeps cached neighbours)
    if m and m.group(1) == kernel and m.group(2) in (None, "<false>", "<false, false>"):
        vals.update(ast.literal_eval(m.group(3)))
avg_ns = calls = None
for r in csv.DictReader(open(os.path.join(d, tag + "_bench_kernel_stats.csv"))):
    if r["Name"].replace("void ", "").startswith(("malio::%s(" % kernel, "malio::%s<false>(" % kernel, "malio::%s<false, false>(" % kernel)):
        avg_ns, calls = float(r["AverageNs"]), int(r["Calls"])
out = {
    "kernel": kernel, "workload": "city3_100k_1M, one launch", "build_id": build_id,
    "FETCH_SIZE_KB": vals.get("FETCH_SIZE"), "WRITE_SIZE_KB": vals.get("WRITE_SIZE"),
    "TCC_EA0_RDREQ": vals.get("TCC_EA0_RDREQ_sum"), "TCC_HIT": vals.get("TCC_HIT_sum"), "TCC_REQ": vals.get("TCC_REQ_sum"),
    "correction": "MI355X_MICROARCH.md (HBM): gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B, half the bytes of wide (16 B/lane) "
                  "coalesced reads -> doubled (upper bound); WRITE_SIZE taken as is",
    "traffic_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
    "rocprof_kernel_ms": avg_ns * 1e-6 if avg_ns else None,
    "rocprof_source": "profiles/" + rnd + "/%s_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py "
                      "--no-cpu-baseline`, %s calls)" % (tag, calls),
    "source": "profiles/" + rnd + "/%s_pmc_summary.txt (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE, "
              "separate passes, tools/pmc_run.sh via tools/profile_round.sh %s)" % (tag, tag),
}
json.dump(out, open(os.path.join(d, tag + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out))
