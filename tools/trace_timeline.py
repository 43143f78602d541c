"""Timeline of ONE blocking step from a rocprofv3 --kernel-trace CSV of tools/blocking_trace.py: every dispatch of the last step
between the profile marks with its duration and the gap to the dispatch before it (us).
    python3 tools/trace_timeline.py <run_kernel_trace.csv> [steps between the marks]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort()
marks = [i for i, r in enumerate(rows) if "profile_mark_kernel" in r[2]]
rows = rows[marks[0] + 1:marks[-1]]
per = len(rows) // steps
rows = rows[-per:]
t0 = rows[0][0]; prev = None; busy = 0
for s, e, n in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    busy += e - s
    print("%8.2f  gap %6.2f  dur %7.2f  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, n[:110]))
    prev = e
print("step: %d dispatches, %.1f us first start -> last end, %.1f us inside kernels" % (len(rows), (rows[-1][1] - t0) / 1e3, busy / 1e3))
