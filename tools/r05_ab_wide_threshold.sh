#!/bin/bash
# NOTE: the WX_* environment knobs used here exist in LAB builds only (python tools/build_lab.py env WX_LAB_ENV; then
# run with the lab library: whisperx_mlx_amd._lib.LIB_PATH / tools/ab_lib.py).  The measurements in profiles/r05_ab_*.txt were
# taken while the knobs were still compiled into the round's working library.
O=gpurun_out
for steps in 12 9 6; do
for env in "WX_WIDE_GEMV_FROM_17=1" ""; do
  env $env timeout -k 10 300 python bench.py --steps $steps --warmup 3 --no-cpu-baseline --no-extra --no-align 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$env'.ljust(24), 'steps $steps', 'value', d['value'], 'ms/step', d['ms_per_step'], 'plan', d['config']['rows_per_pass'], 'x', d['config']['passes_in_flight_per_gpu'], 'live us', d.get('roofline', {}).get('avg_launch_us'))
" | tee -a $O/r05_ab_wide_threshold.txt
done
done
