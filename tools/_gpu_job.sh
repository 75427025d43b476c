cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh r03u 5 > gpurun_out/r03u_profile.log 2>&1
timeout 600 python bench.py > gpurun_out/r03u_bench.json 2> gpurun_out/r03u_bench.err
