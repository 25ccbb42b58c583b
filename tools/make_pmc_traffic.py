"""profiles/pmc_traffic.json and profiles/<round>/ summaries from the output of tools/profile_r02.sh.

  python tools/make_pmc_traffic.py gpurun_out/r02p profiles/r02

For every configuration directory (rollout_k50, rollout_k50_notraj, step) it reads
  stats_kernel_stats.csv                      rocprofv3 --kernel-trace --stats (average duration of the dominant kernel)
  pmc_<COUNTERS>_counter_collection.csv       separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU | SQ wave cycles | ...)
and writes per-launch averages of the dominant kernel.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the gfx950
read-side correction of MI355X_MICROARCH.md applied in full, an upper bound for this mixed-width access pattern
(profiles/README.md).  The file carries the sha256 of the kernel sources it was measured on (bench.py refuses a stale
one) and the VALU issue peak for this kernel's instruction mix, from tools/valu_probe.py + tools/isa_mix.py.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ewn_gym_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def full_launches(v):
    """keep the dispatches of the full K-step launches: bench.py's clock warm-up (unless --no-spin) and its first calibration call launch
    the SAME kernel with fewer steps, and an average over all dispatches would mix them in (r03: 7 of 28 dispatches at a fifth of the
    counter value).  Reference = the median of the upper half; kept = within 15 % of it."""
    if not v:
        return v
    top = sorted(v)[len(v) // 2:]
    ref = top[len(top) // 2]
    keep = [x for x in v if 0.85 * ref <= x <= 1.15 * ref]
    return keep or v


def dominant(rows, want, filt=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if want in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        return None, {}
    k = max(acc, key=lambda k: sum(len(v) for v in acc[k].values()))
    out = {}
    for c, v in acc[k].items():
        # the selection is made on a counter that scales with the work; counters that do not (SQ_WAVES) ride along by index
        keep = full_launches(v) if filt else v
        out[c] = sum(keep) / len(keep)
    return k, out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    from bench import pmc_key
    import subprocess
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"]).decode().strip()
    except Exception:
        head = None
    out = {"source_hash": source_hash(), "git_head_when_summarised": head}
    mixp = os.path.join(dst, "isa_mix.json")
    probe = os.path.join(dst, "valu_probe.json")
    # (directory, dominant kernel, bench.py's key, env steps per launch of that kernel, lanes, isa_mix kernel substring for its own issue peak)
    cfgs = (("rollout_k50", "k_rollout", pmc_key("rollout", "minimax", 3, "philox", 5, 65536, 50, True), 50, 65536, None),
            ("rollout_k20", "k_rollout", pmc_key("rollout", "minimax", 3, "philox", 5, 65536, 20, True), 20, 65536, None),
            ("rollout_k50_notraj", "k_rollout", pmc_key("rollout", "minimax", 3, "philox", 5, 65536, 50, False), 50, 65536, None),
            ("step", "k_step_d3", pmc_key("step", "minimax", 3, "philox", 5, 65536, 1, True), 1, 65536, None),
            # round 3: the max_depth 5 rollout and the flat Monte-Carlo step (config 5's per-GPU shape) carry their own counters and issue peak
            ("d5_k50", "k_rollout", pmc_key("rollout", "minimax", 5, "philox", 5, 65536, 50, True), 50, 65536, "k_rollout_slotsILi5ELi2ELi2ELi1ELb0E"),
            ("mcts7", "k_mcts_rollout_lean", pmc_key("step", "mcts", 3, "philox", 7, 32768, 1, True), 1, 32768, "k_mcts_rollout_lean"))
    mix_all = json.load(open(mixp)) if os.path.exists(mixp) else {"kernels": {}}
    for name, kern, key, K, lanes, mixsub in cfgs:
        d = os.path.join(src, name)
        if not os.path.isdir(d):
            continue
        ent = {"lanes": lanes, "env_steps_per_launch": K}
        if mixsub:
            for kname, kv in mix_all["kernels"].items():
                if mixsub in kname and "peak_wave_insts_per_s" in kv:
                    ent["valu_issue_peak_per_s"] = kv["peak_wave_insts_per_s"]["hot_loop"]
                    ent["valu_issue_peak_source"] = "tools/isa_mix.py hot loop of %s" % kname
        st = os.path.join(d, "stats_kernel_stats.csv")
        if os.path.exists(st):
            for r in csv.DictReader(open(st)):
                if kern in r["Name"]:
                    ent["kernel"] = r["Name"]
                    ent["rocprof_avg_us"] = float(r["AverageNs"]) / 1e3
                    ent["rocprof_calls"] = int(r["Calls"])
                    break
            # the per-dispatch durations, full launches only (see full_launches): what `--stats` averages over ALL dispatches of the name
            tr = os.path.join(d, "stats_kernel_trace.csv")
            if os.path.exists(tr):
                dur = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr)) if kern in r["Kernel_Name"]]
                keep = full_launches(dur) if K > 1 else dur      # one-step kernels (the playout launches vary 1.0-2.5 ms with the game stage): all
                if keep:
                    ent["rocprof_avg_us_all_dispatches"] = ent.get("rocprof_avg_us")
                    ent["rocprof_avg_us"] = sum(keep) / len(keep)
                    ent["rocprof_calls"] = len(keep)
                    ent["rocprof_calls_all_dispatches"] = len(dur)
            shutil.copy(st, os.path.join(dst, "%s_kernel_stats.csv" % name))
        vals = {}
        for f in glob.glob(os.path.join(d, "pmc_*_counter_collection.csv")):
            _, v = dominant(list(csv.DictReader(open(f))), kern, K > 1)
            vals.update(v)
            # keep only the dominant kernel's rows of each pass: the evidence, without the framework's fill kernels
            rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
            if rows:
                with open(os.path.join(dst, "%s_%s" % (name, os.path.basename(f))), "w", newline="") as g:
                    w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
                    w.writeheader()
                    w.writerows(rows[:60])
        for f in ("bench.json", "bench_under_rocprof.json"):
            p = os.path.join(d, f)
            if os.path.exists(p):
                lines = [l for l in open(p) if l.startswith("{")]
                if lines:
                    open(os.path.join(dst, "%s_%s" % (name, f)), "w").write(lines[-1])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            ent["FETCH_SIZE_KB_raw"], ent["WRITE_SIZE_KB_raw"] = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
            ent["hbm_bytes_per_launch"] = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
            ent["per_lane_step_fetch_raw_B"] = vals["FETCH_SIZE"] * 1024 / lanes / K
            ent["per_lane_step_write_raw_B"] = vals["WRITE_SIZE"] * 1024 / lanes / K
        if "SQ_INSTS_VALU" in vals:
            ent["valu_wave_insts_per_launch"] = vals["SQ_INSTS_VALU"]
            ent["valu_wave_insts_per_lane_step"] = vals["SQ_INSTS_VALU"] / lanes / K
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_LDS", "SQ_WAVES"):
            if c in vals:
                ent[c] = vals[c]
        if "SQ_WAVE_CYCLES" in vals and vals["SQ_WAVE_CYCLES"]:
            ent["wait_any_frac"] = vals.get("SQ_WAIT_ANY", 0.0) / vals["SQ_WAVE_CYCLES"]
            ent["active_valu_frac"] = vals.get("SQ_ACTIVE_INST_VALU", 0.0) / vals["SQ_WAVE_CYCLES"]
        out[key] = ent
    if os.path.exists(mixp):
        mix = json.load(open(mixp))
        # the peak the dominant (rollout) kernel's instruction mix could issue at, chip-wide
        for kname, kv in mix["kernels"].items():
            if "k_rollout_slotsILi5ELi2ELi0ELi1ELb0ELi1E" in kname and "peak_wave_insts_per_s" in kv:  # the headline instance (record-only trajectory)
                out["valu_issue_peak_per_s"] = kv["peak_wave_insts_per_s"]["hot_loop"]
                out["valu_issue_peak_source"] = "tools/isa_mix.py hot loop of %s priced with tools/valu_probe.py (valu_probe.json beside it)" % kname
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    if not os.path.exists(probe):
        print("note: %s missing -- run tools/valu_probe.py on the GPU box and tools/isa_mix.py here" % probe, file=sys.stderr)


if __name__ == "__main__":
    main()
