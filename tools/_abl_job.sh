#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 2; do
cd /tmp && export TMPDIR=/tmp
AB_ABLATE_MOMENTS=$v rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl -o tl -- python $GRAFT_REPO_ROOT/tools/time_register.py > $GRAFT_REPO_ROOT/gpurun_out/tl_run_$v.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/register_timeline.py $(find gpurun_out/tl -name "*.db" | head -1) > gpurun_out/s11_timeline_$v.txt 2>&1
rm -rf gpurun_out/tl
done
