"""Summarise one tools/profile_cfg.sh output directory into profiles/<round>/<name>/ (kernel stats, the dominant kernel's rows of
every PMC pass, the bench lines, summary.json with per-launch averages over the full-length launches).

  python tools/collect_cfg_profile.py gpurun_out/r03w/d5_k50 profiles/r02/d5 [kernel-substring] [steps-per-launch]"""
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "k_rollout"
    K = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(dst + "/*"):
        os.remove(f)
    out = {}
    for f in sorted(glob.glob(src + "/pmc_*_counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        acc = {}
        for r in rows:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():   # the warm-up launch is shorter: average the full-length launches only
            m = max(v)
            vv = [x for x in v if x > 0.5 * m]
            out[k] = sum(vv) / len(vv)
            out[k + "_launches"] = len(vv)
        if rows:
            with open(os.path.join(dst, os.path.basename(f)), "w", newline="") as g:
                w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows[:60])
    shutil.copy(src + "/stats_kernel_stats.csv", dst + "/kernel_stats.csv")
    for r in csv.DictReader(open(src + "/stats_kernel_stats.csv")):
        if kern in r["Name"]:
            out["kernel"], out["rocprof_avg_us"], out["rocprof_calls"] = r["Name"], float(r["AverageNs"]) / 1e3, int(r["Calls"])
            break
    out["launch_ns"] = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(src + "/stats_kernel_trace.csv"))
                        if kern in r["Kernel_Name"]]
    for f in ("bench.json", "bench_under_rocprof.json"):
        lines = [x for x in open(os.path.join(src, f)) if x.startswith("{")]
        open(os.path.join(dst, f), "w").write(lines[-1])
    b = json.loads([x for x in open(src + "/bench.json") if x.startswith("{")][-1])
    out["bench_us_per_step"] = b["ms_per_step"] * 1e3
    if out.get("SQ_WAVE_CYCLES"):
        out["valu_pipe_busy_frac_2_waves_per_simd"] = 2 * out["SQ_ACTIVE_INST_VALU"] / out["SQ_WAVE_CYCLES"]
    if out.get("SQ_INSTS_VALU") and out.get("SQ_WAVES"):
        out["valu_wave_insts_per_wave_step"] = out["SQ_INSTS_VALU"] / out["SQ_WAVES"] / K
    json.dump(out, open(dst + "/summary.json", "w"), indent=1)
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if not k.endswith("_launches")}))


if __name__ == "__main__":
    main()
