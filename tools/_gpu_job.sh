cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03x_votes_kernels.txt
cd /tmp && export TMPDIR=/tmp
for K in 1 4; do
  rm -rf /tmp/pv$K
  AB_VOTE_COPIES=$K timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv$K -o t -- python $GRAFT_REPO_ROOT/tools/time_register.py > /tmp/pv$K.log 2>&1 < /dev/null
  echo "== K=$K" >> $GRAFT_REPO_ROOT/gpurun_out/r03x_votes_kernels.txt
  f=$(find /tmp/pv$K -name "*.db" | head -1)
  if [ -n "$f" ]; then timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$f" < /dev/null | head -14 | cut -c1-175 >> $GRAFT_REPO_ROOT/gpurun_out/r03x_votes_kernels.txt; else echo "no db" >> $GRAFT_REPO_ROOT/gpurun_out/r03x_votes_kernels.txt; tail -n 5 /tmp/pv$K.log >> $GRAFT_REPO_ROOT/gpurun_out/r03x_votes_kernels.txt; fi
done
