#!/bin/bash
# PMC passes over the bench (separate runs, counters only + kernel trace), summarised per kernel.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --timing-level 0"
run() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o p -- $CMD > $OUT/$n.log 2>&1
  echo "pmc pass $n rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python $R/tools/pmc_summary.py $OUT > $R/gpurun_out/pmc_summary.txt 2>&1
cat $R/gpurun_out/pmc_summary.txt
