cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4 5 6; do
timeout 900 python -X faulthandler -m pytest tests/test_gpu_background.py tests/test_gpu_batch.py -m gpu -x -q < /dev/null > gpurun_out/r04n_soak_$i.log 2>&1
echo "run $i rc=$?"; tail -n 2 gpurun_out/r04n_soak_$i.log
done
grep -l "Abort\|fault\|Fatal" gpurun_out/r04n_soak_*.log
