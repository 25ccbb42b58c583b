#!/bin/bash
# A/B of builds of libewn_hip.so on one box: tools/ab_bench.sh [-a "<bench args>"]... "<lib or ->[:ENV=VAL]" ...
# alternates the variants, three rounds, every launch shape given with -a (default: the driver's shape and the default bench)
shapes=()
while [ "$1" = "-a" ]; do shapes+=("$2"); shift; shift; done
[ ${#shapes[@]} -eq 0 ] && shapes=("--steps 20 --warmup 5" "")
for i in 1 2 3; do
  for v in "$@"; do
    lib=${v%%:*}; envs=""; [ "$v" != "$lib" ] && envs=${v#*:}
    for shape in "${shapes[@]}"; do
      ( [ "$lib" != "-" ] && export EWN_HIP_LIB=$lib; [ -n "$envs" ] && export $envs
        python bench.py --no-cpu-baseline --no-extras $shape 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%-44s %-44s value %.4g  us/step %.3f  kernel_ms(event) %.4f' % ('$v', '${shape:-default}', d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_ms']))" )
    done
  done
done
