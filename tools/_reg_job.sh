#!/bin/bash
# Run ON THE GPU BOX: detection tests, three timings of the registration stage, and a kernel timeline of it. $1 = tag
TAG=${1:-x}
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_detect_affine.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/${TAG}_tests.txt 2>&1
for i in 1 2 3; do python tools/time_register.py; done 2>&1 | grep align_pairs > gpurun_out/${TAG}_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tl -o tl -- python $GRAFT_REPO_ROOT/tools/time_register.py > $GRAFT_REPO_ROOT/gpurun_out/tl_run.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/register_timeline.py $(find gpurun_out/tl -name "*.db" | head -1) > gpurun_out/${TAG}_timeline.txt 2>&1
rm -rf gpurun_out/tl
