"""Developer probe: per-launch durations from a rocprofv3 --kernel-trace CSV.

  python3 tools/trace_seq.py <run_kernel_trace.csv> [substring ...]

For every kernel whose name contains one of the substrings (default: every kernel with >= 4 launches) prints the
launch count, min / median / max / sigma in microseconds and the sequence of durations in launch order - which is what
shows whether a bimodal average alternates, drifts, or follows the position of the launch inside a step."""
import csv, statistics, sys

path = sys.argv[1]
subs = sys.argv[2:]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
by = {}
for s, e, n in rows:
    by.setdefault(n, []).append((e - s) / 1e3)
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if subs and not any(x in n for x in subs):
        continue
    if not subs and len(d) < 4:
        continue
    sd = statistics.pstdev(d) if len(d) > 1 else 0.0
    print("%-90s n=%4d sum=%9.1f min=%7.1f med=%7.1f max=%7.1f sd=%6.1f" % (n[:90], len(d), sum(d), min(d), statistics.median(d), max(d), sd))
    if subs:
        print("   " + " ".join("%.0f" % x for x in d[:120]))
