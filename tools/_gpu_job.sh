cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_stack.py -m gpu -x -q -k "two_lanes or deep" < /dev/null > gpurun_out/r04m_pytest.log 2>&1
tail -n 5 gpurun_out/r04m_pytest.log
N_LIST=320,512 timeout 600 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu.ids
