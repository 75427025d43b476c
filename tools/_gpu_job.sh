cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --host-planes --no-cpu-baseline --steps 5 < /dev/null 2>gpurun_out/r03au.err | tail -n 1 > gpurun_out/r03au_host.json
timeout 600 python bench.py --mode rowband --force-sharded --no-cpu-baseline --steps 5 < /dev/null 2>>gpurun_out/r03au.err | tail -n 1 > gpurun_out/r03au_rowband.json
timeout 600 python bench.py --force-sharded --no-cpu-baseline --steps 5 < /dev/null 2>>gpurun_out/r03au.err | tail -n 1 > gpurun_out/r03au_frames.json
