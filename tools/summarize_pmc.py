"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) into the per-kernel
table committed under profiles/.  usage: summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> "<command>" [out.json] > out.txt
Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): counter unit KB; on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (doubled in the 'corrected' column); WRITE_SIZE is exact."""
import collections
import csv
import sys


def per_kernel(path, counter, factor):
    agg = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: tot / n * factor * 1024.0 for k, (n, tot) in agg.items()}     # bytes per dispatch


def table(path, counter, factor):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = ["== %s" % counter, "%-84s %8s %12s %18s" % ("kernel", "calls", "avg KB", "corrected avg MB")]
    for k, (n, tot) in rows[:14]:
        out.append("%-84s %8d %12.1f %18.2f" % (k[:84], n, tot / n, tot / n * factor / 1024.0))
    return "\n".join(out)


def main():
    fetch, write, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    print("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), %s" % cmd)
    print("Counter unit: KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the "
          "bytes of wide coalesced reads -> doubled in the 'corrected' column; WRITE_SIZE is exact.")
    print("Per-kernel averages over ALL dispatches of the run (warm-up, style pass and the timed steps).\n")
    print(table(fetch, "FETCH_SIZE", 2.0))
    print()
    print(table(write, "WRITE_SIZE", 1.0))
    if len(sys.argv) > 4:      # machine-readable twin for bench.py's roofline.traffic
        import json
        f, w = per_kernel(fetch, "FETCH_SIZE", 2.0), per_kernel(write, "WRITE_SIZE", 1.0)
        json.dump({"command": cmd, "unit": "bytes per dispatch (FETCH_SIZE x2 on gfx950 + WRITE_SIZE)",
                   "kernels": {k: {"fetch": f.get(k, 0.0), "write": w.get(k, 0.0)} for k in sorted(set(f) | set(w))}},
                  open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
