#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_all.sh <tag> [A|B|AB]     e.g.  gpurun --timeout 1200 -- 'tools/profile_all.sh r03 A'
# Everything profiles/<round>/ is made from, in two calls of <= 20 minutes:
#   A  rocprofv3 kernel stats + PMC passes of the headline bench in its launch shapes, of the max_depth 5 rollout and of the 7x7 flat
#      Monte-Carlo opponent; the bench lines (default, driver shape, MT19937-compat, MCTS both ways)
#   B  the lane-count / opponent sweep, the evaluation wall times, the soak logs, the A2C trainers (throughput, kernel stats, PMC,
#      accuracy against float64), the MFMA probe
# Afterwards, in the build container:
#   hipcc -S --cuda-device-only ... ewn_rollout_s5.hip / ewn_step_d3.hip / ewn_kernels.hip; cat them > /tmp/both.s
#   python tools/isa_mix.py --asm /tmp/both.s --kernel <names> --probe profiles/<round>/valu_probe.json --out profiles/<round>/isa_mix.json
#   python tools/make_pmc_traffic.py gpurun_out/<tag> profiles/<round>
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-rXX}
P=${2:-AB}
O=gpurun_out/$T
mkdir -p $R/$O
cd $R
if [[ $P == *A* ]]; then
  timeout -k 10 500 $R/tools/profile_r02.sh $O > $R/$O/profile.log 2>&1; tail -1 $R/$O/profile.log
  timeout -k 10 200 $R/tools/profile_cfg.sh $O d5_k50 --max-depth 5 --steps 150 --warmup 50 | tail -1
  timeout -k 10 200 $R/tools/profile_cfg.sh $O mcts7 --opponent mcts --board-size 7 --lanes 32768 --num-simulations 40 --num-env-copies 10 --steps 20 --warmup 4 --mode step | tail -1
  cd $R
  timeout -k 10 200 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2>/dev/null
  # the MT19937-compat one-step path, the MCTS opponent both ways
  for o in minimax random; do timeout -k 10 120 python3 bench.py --rng mt19937 --mode step --opponent $o --steps 1000 --warmup 100 --no-cpu-baseline --no-extras > $O/bench_mt19937_step_$o.json 2>/dev/null; done
  for m in rollout step; do
    timeout -k 10 120 python3 bench.py --opponent mcts --num-simulations 10 --num-env-copies 5 --steps 100 --warmup 10 --steps-per-launch 20 --mode $m --no-cpu-baseline --no-extras > $O/bench_mcts5_$m.json 2>/dev/null
    timeout -k 10 120 python3 bench.py --opponent mcts --board-size 7 --lanes 32768 --num-simulations 40 --num-env-copies 10 --steps 20 --warmup 4 --steps-per-launch 10 --mode $m --no-cpu-baseline --no-extras > $O/bench_mcts7_$m.json 2>/dev/null
  done
  echo part A done
fi
if [[ $P == *B* ]]; then
  timeout -k 10 300 tools/sweep_rollout.sh $O/sweep_rollout.txt > /dev/null 2>&1; tail -3 $O/sweep_rollout.txt
  timeout -k 10 100 python3 tools/eval_time.py > $O/eval_time.txt 2>&1
  for p in "" r02 d5 r03 r03b; do
    timeout -k 10 600 python3 tools/soak_parity.py $p > $O/soak_${p:-step}.log 2>&1; echo rc=$? >> $O/soak_${p:-step}.log; tail -2 $O/soak_${p:-step}.log
  done
  timeout -k 10 900 tools/profile_a2c.sh $O/a2c > /dev/null 2>&1; cat $O/a2c/throughput.txt
  [ -x tools/bin/mfma_probe ] || { mkdir -p tools/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/bin/mfma_probe tools/mfma_probe.hip; }
  timeout -k 10 60 tools/bin/mfma_probe 2000 > $O/mfma_probe.txt 2>&1
  echo part B done
fi
