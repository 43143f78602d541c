"""Per-kernel statistics of a rocprofv3 --kernel-trace CSV restricted to the dispatches BETWEEN the first and the last
cnk::profile_mark_kernel dispatch (the timed steps: no warm-up, no per-utterance style pass - whose k = 31 convolutions run
the same conv_mfma instantiations as the upsamplers and would otherwise be averaged into them).

    python3 tools/marked_stats.py <run_kernel_trace.csv> [steps] > profiles/r3_b64_kernel_stats_blocking.csv
"""
import csv, statistics, sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort()
marks = [i for i, r in enumerate(rows) if "profile_mark_kernel" in r[2]]
if len(marks) >= 2:
    rows = rows[marks[0] + 1:marks[-1]]
by = {}
for s, e, n in rows:
    by.setdefault(n, []).append(e - s)
tot = sum(sum(v) for v in by.values())
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "CallsPerStep", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([n, len(d), "%.2f" % (len(d) / steps), sum(d), "%.1f" % (sum(d) / len(d)), "%.2f" % (100.0 * sum(d) / tot), min(d), max(d),
                "%.1f" % (statistics.pstdev(d) if len(d) > 1 else 0.0)])
