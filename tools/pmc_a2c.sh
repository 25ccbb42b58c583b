#!/bin/bash
# usage (GPU box): tools/pmc_a2c.sh <outdir>  -- separate rocprofv3 --pmc passes over the fused A2C trainer's kernels (MFMA pipe busy
# cycles, wave cycles / waits, instruction counts); tools/a2c_pmc_summary.py turns them into profiles/rNN/a2c/pmc_summary.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/a2c_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES"; do
  tag=$(echo $c | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$tag -- python3 $R/tools/a2c_throughput.py --trainer fused --lanes 65536 --updates 20 > $O/pmc_$tag.txt 2> $O/pmc_$tag.err
done
ls $O | head -20
