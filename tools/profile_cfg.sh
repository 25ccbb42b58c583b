#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_cfg.sh <outdir> <name> <bench args...>
# rocprofv3 kernel stats + separate PMC passes of one bench configuration (the per-configuration part of tools/profile_r02.sh)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; name=$2; shift; shift
mkdir -p $OUT/$name
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extras --no-spin"   # no clock warm-up launches under the profiler: they are the same kernel with fewer steps
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o stats -- python3 $R/bench.py $COMMON "$@" > $OUT/$name/bench_under_rocprof.json 2> $OUT/$name/stats.err
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$name -o pmc_$tag -- python3 $R/bench.py $COMMON "$@" > $OUT/$name/pmc_$tag.json 2> $OUT/$name/pmc_$tag.err
done
python3 $R/bench.py $COMMON "$@" > $OUT/$name/bench.json 2> $OUT/$name/bench.err
echo done $name
