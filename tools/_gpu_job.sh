cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03as.txt
cd /tmp && export TMPDIR=/tmp
for B in 1024 2048 4096; do
rm -rf /tmp/pv
AB_VOTE_BLOCKS=$B timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python $GRAFT_REPO_ROOT/tools/time_register.py > /tmp/pv.log 2>&1 < /dev/null
echo "== blocks $B: $(grep align_pairs /tmp/pv.log | cut -c1-70)" >> $GRAFT_REPO_ROOT/gpurun_out/r03as.txt
f=$(find /tmp/pv -name "*.db" | head -1)
if [ -n "$f" ]; then timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$f" < /dev/null | grep -E "tri_vote" | cut -c1-150 >> $GRAFT_REPO_ROOT/gpurun_out/r03as.txt; fi
done
