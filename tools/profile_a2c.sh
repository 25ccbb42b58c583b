#!/bin/bash
# usage (GPU box, from the repo root): tools/profile_a2c.sh <outdir>   -- throughput of both trainers un-profiled, then rocprofv3 kernel
# stats of the fused trainer at 65 536 lanes (profiles/rNN/a2c/ is copied from this)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/a2c}
mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/a2c_throughput.py --trainer both > $O/throughput.txt 2>&1; cat $O/throughput.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o a2c -- python3 $R/tools/a2c_throughput.py --trainer fused --lanes 65536 --updates 100 > $O/throughput_under_rocprof.txt 2>&1
f=$(find $O/rocprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -12 $O/kernel_stats.csv
# PMC passes (MFMA pipe busy, VALU active, waits, instruction counts) -> pmc_summary.json; accuracy against float64 torch per kernel kind
timeout -k 10 300 $R/tools/pmc_a2c.sh ${1:-gpurun_out/a2c}/pmc > /dev/null 2>&1
cd $R
python3 tools/a2c_pmc_summary.py $O/pmc $O/pmc_summary.json > /dev/null 2>&1
: > $O/accuracy.txt
for k in 3 2 1; do EWN_A2C_KERNEL=$k timeout -k 10 100 python3 tools/a2c_accuracy.py 2>/dev/null | tail -3 >> $O/accuracy.txt; done
for k in 2 1; do EWN_A2C_KERNEL=$k timeout -k 10 100 python3 tools/a2c_throughput.py --trainer fused --lanes 65536 --updates 300 2>/dev/null | sed "s/^/EWN_A2C_KERNEL=$k (f32 MFMA) /" >> $O/throughput.txt; done
cat $O/accuracy.txt
