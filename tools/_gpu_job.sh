cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03ad_sweep.txt
timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 >> gpurun_out/r03ad_sweep.txt
AB_DETECT_CHAIN=1 timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 >> gpurun_out/r03ad_sweep.txt
AB_NO_WARP_STREAM=1 timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 >> gpurun_out/r03ad_sweep.txt
GPU_MAX_HW_QUEUES=8 timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 >> gpurun_out/r03ad_sweep.txt
GPU_MAX_HW_QUEUES=2 timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 >> gpurun_out/r03ad_sweep.txt
timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 >> gpurun_out/r03ad_sweep.txt
