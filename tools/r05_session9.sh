#!/bin/bash
# round 5, session 9: four wide passes in flight against the shipped three, with the one-pass GEMV kernels in (round 3 measured
# 64 x 4 below 64 x 3 with the row-group kernels).  Product library, bench flags only.
set -o pipefail
O=gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-align"
for mode in "" "--rows-per-pass 80 --streams 4" "--rows-per-pass 64 --streams 4" "" "--rows-per-pass 80 --streams 4"; do
  timeout -k 10 240 $B $mode > $O/s9.json 2>$O/s9_err.log || { tail -5 $O/s9_err.log; exit 1; }
  python -c "
import json;d=json.loads(open('$O/s9.json').read().strip().splitlines()[-1]);c=d['config']
print('$mode'.ljust(34),'value',d['value'],'ms/step',d['ms_per_step'],'plan',c['rows_per_pass'],'x',c['passes_in_flight_per_gpu'],'live launch us',d['roofline']['avg_launch_us'],'memory GB',d['gpu_memory_gb']['in_use_after_the_timed_run'])" | tee -a $O/r05_ab_four_wide_passes.txt
done
