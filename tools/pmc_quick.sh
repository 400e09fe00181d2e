#!/bin/bash
# One SQ pass of PMC counters over a short bench; prints the target kernel's numbers.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --timing-level 0"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq1 -o p -- $CMD > $OUT/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $CMD > $OUT/sq2.log 2>&1
tail -3 $OUT/sq2.log
python $R/tools/pmc_summary.py $OUT > $R/gpurun_out/pmc_summary.txt 2>&1
awk "/rowpass|weight_grad_kernel|target_fused/{f=1;n=0} f{print; n++} n>17{f=0}" $R/gpurun_out/pmc_summary.txt
