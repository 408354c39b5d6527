#!/bin/bash
# NOTE: the WX_* environment knobs used here exist in LAB builds only (python tools/build_lab.py env WX_LAB_ENV; then
# run with the lab library: whisperx_mlx_amd._lib.LIB_PATH / tools/ab_lib.py).  The measurements in profiles/r05_ab_*.txt were
# taken while the knobs were still compiled into the round's working library.
# Round 5: one GPU session that verifies everything changed since the last green run and measures the A/Bs.
# Each step writes under gpurun_out/; steps are independent of each other's success except where joined with &&.
mkdir -p gpurun_out
O=gpurun_out
echo "== full GPU suite" ; date
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_t4.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05_t4.log; tail -4 $O/r05_t4.log
echo "== bench (driver command shape: 20 steps)"; date
timeout -k 10 400 python bench.py --steps 20 > $O/r05_b2.json 2> $O/r05_b2.err; echo "bench rc=$?"; python tools/show_bench.py $O/r05_b2.json 2>/dev/null | head -40 || tail -c 1500 $O/r05_b2.json
echo "== wide GEMV A/B"; date
for mode in "--rows-per-pass 64 --streams 1" "--rows-per-pass 112 --streams 1" ""; do
  for env in "WX_NO_WIDE_GEMV=1" ""; do
    env $env timeout -k 10 300 python bench.py --steps 14 --warmup 3 --no-cpu-baseline --no-extra --no-align $mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$env'.ljust(18), '$mode'.ljust(34), 'value', d['value'], 'ms/step', d['ms_per_step'], 'plan', d['config']['rows_per_pass'], 'x', d['config']['passes_in_flight_per_gpu'], 'live us', d.get('roofline', {}).get('avg_launch_us'), 'selfq', d.get('fused_launch_selfq_blocks'))
" >> $O/r05_ab_wide_gemv.txt 2>&1
  done
done
cat $O/r05_ab_wide_gemv.txt
date
