#!/bin/bash
# prologue launch of learn(): the sampler's share against the rebuild's.  The PEARL_AMD_DEBUG_PROLOGUE knob it drove (one repack
# workgroup instead of 48) was a timing experiment and is no longer in the library (record: profiles/r06_m_prologue_sampler.txt).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/prof_pro
  PEARL_AMD_DEBUG_PROLOGUE=$v rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_pro -o sc -- python $R/tools/shortcall.py --trace > /dev/null 2>&1
  DB=$(ls $R/gpurun_out/prof_pro/*.db $R/gpurun_out/prof_pro/*/*.db 2>/dev/null | head -1)
  echo "PEARL_AMD_DEBUG_PROLOGUE=$v"; python $R/tools/rocpd_summary.py $DB | grep -E "prologue" | cut -c1-150
  rm -f $DB
  PEARL_AMD_DEBUG_PROLOGUE=$v python $R/tools/shortcall.py --rounds 1,20 --calls 40 2>/dev/null | cut -c1-120
done
