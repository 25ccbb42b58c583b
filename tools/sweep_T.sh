#!/bin/bash
# usage: tools/sweep_T.sh [bench args]  (GPU box): env steps/s of the lean depth-3 kernel for every lanes-per-game T and lane count
for n in 4096 8192 16384 32768 65536 131072 262144 524288 1048576; do
  for t in 1 2 4; do
    EWN_D3_T=$t python bench.py --lanes $n --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=%8d T=%d  %.4g steps/s  %.2f us' % ($n, $t, d['value'], d['ms_per_step']*1e3))"
  done
done
