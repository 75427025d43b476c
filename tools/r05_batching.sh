#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r05aj_pair.txt; : > $out
for rep in 1 2 3; do
  for cfg in "AB_NOOP=1" "AB_LABEL_SINGLE=1"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
    env $cfg NOALIGN=1 REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-130 $out
