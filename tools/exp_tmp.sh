cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "X=0" "PEARL_AMD_X_FLAG=1"; do
  rm -rf $R/gpurun_out/prof_sc
  env $cfg timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
  DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
  tag=$(echo $cfg | tr -c 'A-Za-z0-9' '_')
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/sc_timeline_$tag.txt 2>&1
  echo "== $cfg"; grep -B3 -A6 "16, 16, 1>" $R/gpurun_out/sc_timeline_$tag.txt | cut -c1-110
  rm -f $DB
done
