#!/bin/bash
# usage: tools/pmc_mcts.sh <tag>   (GPU box, repo root): SQ counters of the MCTS playout kernel on the cfg-B shape
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o $n -- python3 $R/bench.py --opponent mcts --board-size 7 --lanes 32768 --num-simulations 400 --num-env-copies 1 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/pmc_$tag/$n.json 2> $R/gpurun_out/pmc_$tag/$n.err
done
