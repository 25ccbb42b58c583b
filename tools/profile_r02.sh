#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_r02.sh <outdir>
# rocprofv3 kernel stats + separate PMC passes (TCC slot budget, MI355X_MICROARCH.md "rocprofv3 PMC slots") of the headline bench
# in its three launch shapes.  Output goes under <outdir> (gpurun_out/...); tools/make_pmc_traffic.py turns it into
# profiles/pmc_traffic.json and the summaries to commit under profiles/r02/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/r02p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-extras --no-spin"   # no clock warm-up launches under the profiler: they are the same kernel with fewer steps
run_cfg() { # name, bench args...
  local name=$1; shift
  mkdir -p $OUT/$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o stats -- python3 $R/bench.py $COMMON --steps 1000 --warmup 100 "$@" > $OUT/$name/bench_under_rocprof.json 2> $OUT/$name/stats.err
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
    tag=$(echo $c | tr ' ' '+')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$name -o pmc_$tag -- python3 $R/bench.py $COMMON --steps 200 --warmup 20 "$@" > $OUT/$name/pmc_$tag.json 2> $OUT/$name/pmc_$tag.err
  done
  python3 $R/bench.py $COMMON "$@" > $OUT/$name/bench.json 2> $OUT/$name/bench.err
}
run_cfg rollout_k50 --steps-per-launch 50
run_cfg rollout_k20 --steps-per-launch 20     # the launch shape of `bench.py --steps 20` (the driver's invocation)
run_cfg rollout_k50_notraj --steps-per-launch 50 --no-trajectory
run_cfg step --mode step
echo done
