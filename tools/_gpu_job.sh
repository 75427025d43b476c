cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -n 8 > gpurun_out/r03v_pytest_gpu.log
timeout 1500 bash tools/profile_bench.sh r03v 5 > gpurun_out/r03v_profile.log 2>&1 < /dev/null
timeout 600 python bench.py > gpurun_out/r03v_bench.json 2> gpurun_out/r03v_bench.err < /dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03v_smoke.txt 2>&1 < /dev/null
