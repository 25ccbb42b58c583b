#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <bench args...>   (run on the GPU box from the repo root)
# Separate passes for FETCH_SIZE and WRITE_SIZE (TCC slot budget, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
mkdir -p $R/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$tag -o $c -- python3 $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/pmc_$tag/$c.json 2> $R/gpurun_out/pmc_$tag/$c.err
done
