"""profiles/pmc_traffic.json from the outputs of tools/pmc_passes.sh (gpurun_out/pmc_<tag>_{minimax,random,mt}/): per-launch
FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU of the lean step kernel at 65 536 lanes.  usage: python tools/make_pmc_traffic.py <tag>"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = {}
for cfg, key in (("minimax", "minimax_d3_philox"), ("random", "random_d3_philox"), ("mt", "minimax_d3_mt19937")):
    d = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, cfg))
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, c + "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if "k_step_d3" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        if acc:
            k = max(acc, key=lambda k: len(acc[k]))
            vals[c] = sum(acc[k]) / len(acc[k])
    if len(vals) < 3:
        continue
    N = 65536
    out[key] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    out[key + "_valu_wave_insts"] = vals["SQ_INSTS_VALU"]
    out[key + "_detail"] = {"lanes": N, "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KB_raw": vals["WRITE_SIZE"],
                            "per_lane_fetch_raw_B": vals["FETCH_SIZE"] * 1024 / N, "per_lane_write_raw_B": vals["WRITE_SIZE"] * 1024 / N,
                            "SQ_INSTS_VALU": vals["SQ_INSTS_VALU"],
                            "note": "per k_step_d3 launch; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 read-side correction applied in full: an upper bound for this mixed-width pattern, profiles/README.md)"}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.endswith("_detail")}, indent=1))
