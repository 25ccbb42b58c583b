#!/bin/bash
# usage (build container, from the repo root): tools/collect_all.sh <tag> [round-dir]
# Turns gpurun_out/<tag> (written by tools/profile_all.sh on the GPU box) into profiles/<round>/ and profiles/pmc_traffic.json:
# device assembly of the headline kernels -> isa_mix.json; PMC passes -> pmc_traffic.json; the max_depth 5 summary; sweep, evaluation
# times, soak logs and the two bench lines.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:?tag}; D=${2:-profiles/r02}
cd $R
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -S --cuda-device-only"
/opt/rocm/bin/hipcc $FLAGS -o /tmp/_roll5.s ewn_gym_amd/csrc/ewn_rollout_s5.hip 2>/dev/null &
/opt/rocm/bin/hipcc $FLAGS -o /tmp/_step.s ewn_gym_amd/csrc/ewn_step_d3.hip 2>/dev/null
wait
/opt/rocm/bin/hipcc $FLAGS -o /tmp/_kern.s ewn_gym_amd/csrc/ewn_kernels.hip 2>/dev/null
cat /tmp/_roll5.s /tmp/_step.s /tmp/_kern.s > /tmp/_both.s
[ -f $D/valu_probe.json ] || cp profiles/r02/valu_probe.json $D/valu_probe.json   # the per-class issue rates (tools/valu_probe.py, measured in round 2: hardware, not code)
python3 tools/isa_mix.py --asm /tmp/_both.s --kernel "k_rollout_slotsILi5ELi2ELi0ELi1ELb0ELi1E,k_rollout_slotsILi5ELi2ELi0ELi1ELb0ELi0E,k_rollout_d3ILi5ELi2ELi0ELi1ELi0E,k_step_d3ILi5ELi2ELi0ELi1E,k_rollout_slotsILi5ELi2ELi2ELi1E,k_mcts_rollout_lean" --out $D/isa_mix.json > /dev/null
python3 tools/make_pmc_traffic.py gpurun_out/$T $D > /dev/null
python3 tools/collect_cfg_profile.py gpurun_out/$T/d5_k50 $D/d5 > /dev/null
[ -d gpurun_out/$T/mcts7 ] && python3 tools/collect_cfg_profile.py gpurun_out/$T/mcts7 $D/mcts7 > /dev/null
O=gpurun_out/$T
for f in sweep_rollout.txt eval_time.txt mfma_probe.txt bench_default.json bench_driver_shape.json bench_mt19937_step_minimax.json bench_mt19937_step_random.json \
         bench_mcts5_rollout.json bench_mcts5_step.json bench_mcts7_rollout.json bench_mcts7_step.json; do
  [ -s $O/$f ] && cp $O/$f $D/$f
done
for p in step mcts predict r02 d5 r03 long; do [ -s $O/soak_$p.log ] && cp $O/soak_$p.log $D/soak_$p.log; done
if [ -d $O/a2c ]; then
  mkdir -p $D/a2c
  for f in throughput.txt throughput_under_rocprof.txt kernel_stats.csv pmc_summary.json accuracy.txt; do [ -s $O/a2c/$f ] && cp $O/a2c/$f $D/a2c/$f; done
fi
python3 - <<PY
import json
pm = json.load(open("profiles/pmc_traffic.json"))
print("source_hash", pm["source_hash"])
for k, v in pm.items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("rocprof_avg_us", "valu_wave_insts_per_lane_step", "active_valu_frac", "hbm_bytes_per_launch")})
for f in ("$D/bench_default.json", "$D/bench_driver_shape.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d["roofline"]
    print(f, "%.4g" % d["value"], "%.3f us/step" % (d["ms_per_step"] * 1e3), "kernel_ms", round(r["kernel_ms"], 4), r["kernel_ms_rocprof"], "hbm frac %.3f" % r["frac"],
          "valu", r["valu_issue"] and round(r["valu_issue"]["frac"], 3), {k: round(v["us_per_step"], 2) for k, v in (d["extras"] or {}).items()})
PY
