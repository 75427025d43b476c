cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r03an_scan.txt
cd /tmp && export TMPDIR=/tmp
for T in time_f_rows time_batch time_fits; do
rm -rf /tmp/ps_$T
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ps_$T -o t -- python $GRAFT_REPO_ROOT/tools/$T.py > /tmp/ps_$T.log 2>&1 < /dev/null
echo "== $T" >> $GRAFT_REPO_ROOT/gpurun_out/r03an_scan.txt
grep -v "amdgpu.ids\|rocprofv3\|^W2026\|^E2026" /tmp/ps_$T.log | tail -n 14 | cut -c1-160 >> $GRAFT_REPO_ROOT/gpurun_out/r03an_scan.txt
f=$(find /tmp/ps_$T -name "*.db" | head -1)
if [ -n "$f" ]; then timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_summary.py "$f" < /dev/null | grep -v "at::native\|rocprim" | head -16 | cut -c1-140 >> $GRAFT_REPO_ROOT/gpurun_out/r03an_scan.txt; fi
done
