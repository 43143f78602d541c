"""Regenerate DESIGN.md's round table (between the round-table markers) from tools/design_round_table.tpl and the committed profiles
(profiles/<round>_*): run after tools/collect_round.sh's output has been copied into profiles/.   python tools/fill_design.py [round]"""
import json, re, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r6"
P = "profiles/%s_" % R


def J(name):
    return json.load(open(P + name + "_bench.json"))


d = J("b64"); r = d["roofline"]; f = J("b64_f32"); fr = f["roofline"]
b1 = J("b1"); b1w = J("b1win"); b128 = J("b128s2"); b128w = J("b128s2win"); m4 = J("b128s2mem4"); comm = J("b64_comm") if False else json.load(open(P + "b64_bench_comm.json"))
pmc = json.load(open(P + "b64_pmc.json"))
st = dict(re.findall(r"(emformer|decoder|vocoder|pipelined)\D+([0-9.]+) ms", open(P + "stage_times.txt").read()))
tests = open(P + "pytest_gpu.txt").read().strip().splitlines()[-1]
k20 = json.load(open(P + "b64_bench_k20.json"))
fp = re.findall(r"fixed_plan=(True|False) ms/step ([0-9.]+) p50 ([0-9.]+) vocoder alone ([0-9.]+)", open(P + "fixed_plan_cost.txt").read())
fpm = lambda flag, i: sorted(float(x[i]) for x in fp if x[0] == flag)[sum(1 for x in fp if x[0] == flag) // 2]      # median of the runs
cpu = d["cpu_baseline"]


def kern(sub):
    for name, k in pmc.get("kernels", {}).items():
        if sub in name:
            return k
    return None


dec = kern("decoder_mega")
fill = {
    "MS64": "%.3f" % d["ms_per_step"], "V64": "%.1f" % (d["value"] / 1e3), "RT64": "%d" % round(d["realtime_streams_supported"], -1),
    "UNP64": "%.3f" % r["ms_per_step_unprimed"], "F32MS": "%.3f" % r["f32_ms_per_step"],
    "P50": "%.3f" % d["p50_latency_ms"], "P95": "%.3f" % r["p95_latency_ms"], "F32P50": "%.2f" % r["f32_p50_latency_ms"],
    "B1": "%.3f" % r["latency_b1_ms"], "B4": "%.3f" % r["latency_b4_ms"], "F32B1": "%.3f" % fr.get("latency_b1_ms", float("nan")),
    "KMS": "%.3f" % r["kernel_ms_per_step"], "ACH": "%.1f" % r["achieved"], "FRAC": "%.3f" % r["frac"],
    "F128": "%.2f" % r["dominant_frac_c128"], "F64": "%.2f" % r["dominant_frac_c64"], "F32C": "%.2f" % r["dominant_frac_c32"],
    "F32FRAC": "%.3f" % r["f32_frac"],
    "EMF": st.get("emformer", "?"), "DEC": st.get("decoder", "?"), "VOC": st.get("vocoder", "?"), "PIPE": st.get("pipelined", "?"),
    "FE": "%.3f" % r["frontend_cost_ms"], "STF": "%.0f" % r["step_tflops"], "SMF": "%.2f" % r["step_mfma_frac"],
    "HBM": "%.2f" % (r["hbm_bytes_per_step_counter"] / 1e9),
    "FETCH": "%.2f" % (pmc.get("fetch_bytes_per_step", 0) / 1e9) if pmc.get("fetch_bytes_per_step") else "?",
    "WRITE": "%.2f" % (pmc.get("write_bytes_per_step", 0) / 1e9) if pmc.get("write_bytes_per_step") else "?",
    "DECF": "%.0f" % (dec["fetch"] / dec["dispatches_per_step"] / 1e6) if dec else "?",
    "B1WIN": "%.3f" % b1w["ms_per_step"], "B128MS": "%.3f" % b128["ms_per_step"], "B128V": "%.1f" % (b128["value"] / 1e3),
    "B128P50": "%.2f" % b128["p50_latency_ms"], "B128WIN": "%.2f" % b128w["ms_per_step"], "MEM4": "%.3f" % m4["ms_per_step"],
    "COMM": "%.3f" % comm["ms_per_step"], "K20": "%.3f" % k20["ms_per_step"], "VOCA": "%.3f" % r["vocoder_alone_ms"],
    "FPD": "%.4f" % fpm("False", 1), "FPF": "%.4f" % fpm("True", 1), "FPPD": "%.3f" % fpm("False", 2), "FPPF": "%.3f" % fpm("True", 2),
    "FPVD": "%.3f" % fpm("False", 3), "FPVF": "%.3f" % fpm("True", 3), "CPU": "%.1f" % cpu["value"], "CPUS": "%.1f" % cpu["stateful_value"], "TESTS": tests.strip("= "),
}
t = open("tools/design_round_table.tpl").read()
for k, v in fill.items():
    t = t.replace("@" + k + "@", str(v))
left = re.findall(r"@[A-Z0-9]+@", t)
s = open("DESIGN.md").read()
a = s.index("<!-- round-table-begin")
a = s.index("\n", a) + 1
b = s.index("<!-- round-table-end -->")
open("DESIGN.md", "w").write(s[:a] + t + s[b:])
print("table regenerated; placeholders left:", sorted(set(left)))
