cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q -k "host_frames" < /dev/null > gpurun_out/r04t_pytest.log 2>&1
tail -n 25 gpurun_out/r04t_pytest.log
