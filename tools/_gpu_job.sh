cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh r04final 7 > gpurun_out/r04final_profile.log 2>&1
tail -3 gpurun_out/r04final_profile.log
cp gpurun_out/prof_r04final/stack_pmc.json profiles/stack_pmc.json 2>/dev/null
timeout 600 python bench.py > gpurun_out/r04final_bench.json 2> gpurun_out/r04final_bench.err < /dev/null
for c in C1 C3 C5; do timeout 900 python bench.py --config $c > gpurun_out/r04final_bench_$c.json 2> gpurun_out/r04final_bench_$c.err < /dev/null; done
timeout 600 python bench.py --host-planes --no-cpu-baseline > gpurun_out/r04final_bench_host.json 2> gpurun_out/r04final_bench_host.err < /dev/null
timeout 600 python bench.py --force-sharded --no-cpu-baseline > gpurun_out/r04final_bench_sharded.json 2> /dev/null < /dev/null
timeout 600 python bench.py --force-sharded --mode rowband --no-cpu-baseline > gpurun_out/r04final_bench_rowband.json 2> /dev/null < /dev/null
timeout 600 python tools/time_batch.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04final_batch.txt
ls gpurun_out/prof_r04final
