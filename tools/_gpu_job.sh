cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_background.py tests/test_gpu_masked.py tests/test_gpu_phasecorr.py tests/test_gpu_compose.py tests/test_gpu_extras.py tests/test_gpu_spcc.py tests/test_gpu_full_size.py -m gpu -x -q < /dev/null 2>&1 | tail -n 6 > gpurun_out/r03ac_pytest.log
timeout 300 python tools/time_c5.py < /dev/null > gpurun_out/r03ac_c5.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pc5 -o t -- python $GRAFT_REPO_ROOT/tools/time_c5.py > /tmp/pc5.log 2>&1 < /dev/null
f=$(find /tmp/pc5 -name "*.db" | head -1)
if [ -n "$f" ]; then timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$f" < /dev/null | head -12 | cut -c1-150 >> $GRAFT_REPO_ROOT/gpurun_out/r03ac_c5.txt; fi
