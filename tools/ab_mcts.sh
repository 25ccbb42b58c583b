#!/bin/bash
# usage: tools/ab_mcts.sh  (GPU box): MCTS bench for the default build and every ewn_gym_amd/lib/variants/*.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python bench.py --opponent mcts --board-size 7 --lanes 32768 --num-simulations 400 --num-env-copies 1 --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '7x7/400', d['ms_per_step'])"
        python bench.py --opponent mcts --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '5x5/10x5', d['ms_per_step'])"; }
run default
for v in $R/ewn_gym_amd/lib/variants/*.so; do [ -e "$v" ] && EWN_HIP_LIB=$v run $(basename $v); done
exit 0
