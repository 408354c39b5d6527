#!/bin/bash
# NOTE: the WX_* environment knobs used here exist in LAB builds only (python tools/build_lab.py env WX_LAB_ENV; then
# run with the lab library: whisperx_mlx_amd._lib.LIB_PATH / tools/ab_lib.py).  The measurements in profiles/r05_ab_*.txt were
# taken while the knobs were still compiled into the round's working library.
# Round 5, GPU session 4: the one-pass GEMV at the driver's plan (112 + 112 + 96 x 3 in flight), on / off, twice each;
# config 4's alignment stage with this round's forward cuts against round 4's.
mkdir -p gpurun_out
O=gpurun_out
for rep in 1 2; do
for env in "" "WX_NO_WIDE_GEMV=1"; do
  env $env timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-align 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$env'.ljust(18), 'value', d['value'], 'ms/step', d['ms_per_step'], 'plan', d['config']['rows_per_pass'], 'x', d['config']['passes_in_flight_per_gpu'], 'live us', d.get('roofline', {}).get('avg_launch_us'), 'selfq', d.get('fused_launch_selfq_blocks'))
" | tee -a $O/r05_ab_wide_gemv_driver.txt
done
done
timeout -k 10 300 python tools/ab_align_cuts.py 2>&1 | grep -v "^Failed to align" | tee $O/r05_ab_align_cuts.txt | head -60
