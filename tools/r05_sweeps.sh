#!/bin/bash
# NOTE: the WX_* environment knobs used here exist in LAB builds only (python tools/build_lab.py env WX_LAB_ENV; then
# run with the lab library: whisperx_mlx_amd._lib.LIB_PATH / tools/ab_lib.py).  The measurements in profiles/r05_ab_*.txt were
# taken while the knobs were still compiled into the round's working library.
# Round 5, second GPU session: the poll-window sweep of the fused decode launch and the GEMM lab knobs.
mkdir -p gpurun_out
O=gpurun_out
echo "== DL_POLL sweep on the driver job"; date
for polls in 0 8 32 128 512; do
  env WX_DL_POLL=$polls timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-align 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d.get('roofline', {})
print('WX_DL_POLL=$polls'.ljust(16), 'value', d['value'], 'ms/step', d['ms_per_step'], 'live launch us', r.get('avg_launch_us'), 'frac', r.get('frac'), 'selfq blocks', d.get('fused_launch_selfq_blocks'))
" >> $O/r05_ab_dl_poll.txt 2>&1
done
cat $O/r05_ab_dl_poll.txt
echo "== GEMM knobs"; date
for cfg in "" "WX_GEMM_STAGGER_US=8" "WX_GEMM_STAGGER_US=20" "WX_GEMM_STAGGER_US=40" "WX_GEMM_NT=1" "WX_GEMM_NT=1 WX_GEMM_STAGGER_US=20"; do
  env AB_PIPE_ONLY=1 $cfg timeout -k 10 200 python tools/ab_gemm_pipe.py 2>&1 | grep rows >> $O/r05_ab_gemm_knobs.txt
done
cat $O/r05_ab_gemm_knobs.txt
date
