"""Per-launch averages of the fused A2C trainer's kernels from tools/pmc_a2c.sh output.
  python tools/a2c_pmc_summary.py gpurun_out/<dir> profiles/rNN/a2c/pmc_summary.json"""
import collections
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith(("void k_a2c", "k_a2c", "void k_rollout_mlp")):
            acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    if e.get("GRBM_GUI_ACTIVE"):   # GRBM_GUI_ACTIVE comes summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES summed over the 1 024 SIMDs (cycles)
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_pipe_busy_frac"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0)
    if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_ACTIVE_INST_VALU"):   # quad-cycles summed over SIMDs; includes the MFMA issue cycles
        e["valu_active_frac"] = e["SQ_ACTIVE_INST_VALU"] * 4.0 / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if e.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac_of_wave_cycles"] = e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
        e["wait_inst_any_frac_of_wave_cycles"] = e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    out[k] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
