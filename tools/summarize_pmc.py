"""Summarise separate rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) of
`bench.py --marks` into the per-kernel table committed under profiles/.

    summarize_pmc.py --cmd "<command>" --steps K --out profiles/r2_b64_pmc.json  pass1.csv pass2.csv ... > profiles/r2_b64_pmc.txt

Only dispatches BETWEEN the two cnk::profile_mark_kernel dispatches of a pass are kept (the timed steps: no warm-up, no
style pass).  Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (doubled here), WRITE_SIZE is
exact; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs:
mfma_busy_frac = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)."""
import argparse
import collections
import csv
import json

MARK = "profile_mark_kernel"


def timed_rows(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = sorted({int(r["Dispatch_Id"]) for r in rows if MARK in r["Kernel_Name"]})
    if len(marks) >= 2:
        lo, hi = marks[0], marks[-1]
        rows = [r for r in rows if lo < int(r["Dispatch_Id"]) < hi]
        return rows, True
    return [r for r in rows if MARK not in r["Kernel_Name"]], False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cmd", default="")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("passes", nargs="+")
    a = ap.parse_args()
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))   # kernel -> counter -> [dispatches, sum]
    marked = True
    for p in a.passes:
        rows, ok = timed_rows(p)
        marked = marked and ok
        for r in rows:
            c = per[r["Kernel_Name"]][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    kernels = {}
    tot_fetch = tot_write = 0.0
    for k, cs in per.items():
        e = {"dispatches_per_step": max(v[0] for v in cs.values()) / a.steps}
        if "FETCH_SIZE" in cs:
            e["fetch"] = cs["FETCH_SIZE"][1] / cs["FETCH_SIZE"][0] * 2.0 * 1024.0
            tot_fetch += cs["FETCH_SIZE"][1] * 2.0 * 1024.0
        if "WRITE_SIZE" in cs:
            e["write"] = cs["WRITE_SIZE"][1] / cs["WRITE_SIZE"][0] * 1024.0
            tot_write += cs["WRITE_SIZE"][1] * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs and cs["GRBM_GUI_ACTIVE"][1] > 0:
            e["mfma_busy_cycles"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1] / cs["SQ_VALU_MFMA_BUSY_CYCLES"][0]
            e["gui_active_cycles_per_xcd"] = cs["GRBM_GUI_ACTIVE"][1] / cs["GRBM_GUI_ACTIVE"][0] / 8.0
            e["mfma_busy_frac"] = e["mfma_busy_cycles"] / (e["gui_active_cycles_per_xcd"] * 256 * 4)
        if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
            h, m = cs["TCC_HIT_sum"][1], cs["TCC_MISS_sum"][1]
            e["l2_hit_rate"] = h / (h + m) if h + m > 0 else None
        e.setdefault("fetch", 0.0); e.setdefault("write", 0.0)
        kernels[k] = e
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conan_amd", "libconan_hip.so")
    try:
        sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()      # the library the passes ran with (bench.py compares it with the one it loads)
    except OSError:
        sha = None
    out = {"command": a.cmd, "steps": a.steps, "filtered_to_marked_steps": marked, "lib_sha256": sha,
           "unit": "bytes per dispatch (FETCH_SIZE x2 on gfx950, WRITE_SIZE exact), averaged over the dispatches of the timed steps",
           "bytes_per_step": (tot_fetch + tot_write) / a.steps, "fetch_bytes_per_step": tot_fetch / a.steps, "write_bytes_per_step": tot_write / a.steps,
           "kernels": kernels}
    print("rocprofv3 --pmc passes (separate runs, --kernel-trace only) of: %s" % a.cmd)
    print("dispatches between the two cnk::profile_mark_kernel marks only: %s; %d timed steps" % (marked, a.steps))
    print("HBM bytes per step (all kernels): fetch %.1f MB (FETCH_SIZE x2, gfx950) + write %.1f MB = %.1f MB\n" % (tot_fetch / a.steps / 1e6, tot_write / a.steps / 1e6, (tot_fetch + tot_write) / a.steps / 1e6))
    print("%-70s %7s %10s %10s %10s %8s" % ("kernel", "n/step", "fetch MB", "write MB", "mfma busy", "L2 hit"))
    for k, e in sorted(kernels.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"]) * kv[1]["dispatches_per_step"]):
        print("%-70s %7.1f %10.2f %10.2f %10s %8s" % (k[:70], e["dispatches_per_step"], e["fetch"] / 1e6, e["write"] / 1e6,
                                                      ("%.3f" % e["mfma_busy_frac"]) if "mfma_busy_frac" in e else "-",
                                                      ("%.3f" % e["l2_hit_rate"]) if e.get("l2_hit_rate") is not None else "-"))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
