#!/bin/bash
# usage (GPU box): tools/sweep_rollout.sh <outfile>   -- ewn_step_k at several lane counts and lanes per game (EWN_ROLLOUT_T)
out=${1:-gpurun_out/sweep_rollout.txt}
: > $out
for N in 16384 32768 65536 131072 262144 1048576; do
  for T in 1 2; do
    EWN_ROLLOUT_T=$T python3 bench.py --no-cpu-baseline --no-extras --lanes $N --steps 400 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('N=$N T=$T traj   %.4g steps/s  %.2f us/step' % (d['value'], d['ms_per_step']*1e3))" >> $out
    EWN_ROLLOUT_T=$T python3 bench.py --no-cpu-baseline --no-extras --no-trajectory --lanes $N --steps 400 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('N=$N T=$T notraj %.4g steps/s  %.2f us/step' % (d['value'], d['ms_per_step']*1e3))" >> $out
  done
done
for S in 6 7 8; do
  python3 bench.py --no-cpu-baseline --no-extras --board-size $S --steps 400 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('S=$S N=65536 traj %.4g steps/s  %.2f us/step' % (d['value'], d['ms_per_step']*1e3))" >> $out
done
for opp in "--opponent random" "--max-depth 1" "--max-depth 2" "--max-depth 4" "--max-depth 5 --steps 100 --warmup 10" "--heuristic min_dist" "--heuristic attk" "--heuristic two_min_dist" "--opponent mcts --num-simulations 10 --num-env-copies 5 --steps 100 --warmup 10"; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 50 $opp 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$opp: %.4g steps/s  %.2f us/step  (%s)' % (d['value'], d['ms_per_step']*1e3, d['config']['mode']))" >> $out
done
cat $out
