cd /tmp && export TMPDIR=/tmp
OUT=/tmp/sq_out; rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace -d $OUT -o a -- python tools/time_stack_bench_data.py > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o b -- python tools/time_stack_bench_data.py > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY --kernel-trace -d $OUT -o c -- python tools/time_stack_bench_data.py > $OUT/c.log 2>&1
for f in a b c; do db=$(ls $OUT/*/${f}_results.db $OUT/${f}_results.db 2>/dev/null | head -1); echo "== $f $db"; [ -n "$db" ] && timeout 60 python tools/rocpd_pmc.py "$db" | grep -i "stack_sigma\|counter" | head -12; done
tail -3 $OUT/a.log
