cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_stack.py -m gpu -x -q < /dev/null > gpurun_out/r04h_pytest.log 2>&1
tail -n 12 gpurun_out/r04h_pytest.log
N_LIST=320,512 timeout 600 python tools/time_stack_deep.py > gpurun_out/r04h_deep.txt 2>&1
cat gpurun_out/r04h_deep.txt
