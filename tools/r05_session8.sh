#!/bin/bash
# round 5, session 8: encoder attention at three blocks per CU (shipped: 168 registers + 64 B of scratch) against two (188 registers, no
# scratch) IN THE JOB -- value, the encode stage and the bench's attention probe, alternating libraries on one box
set -o pipefail
O=gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-align"
cp whisperx_mlx_amd/libwxhip.so /tmp/occ3.so
for r in 1 2; do
  for v in occ3 occ2; do
    if [ $v = occ3 ]; then cp /tmp/occ3.so whisperx_mlx_amd/libwxhip.so; else cp tools/_occ2_lib.bin whisperx_mlx_amd/libwxhip.so; fi
    timeout -k 10 200 $B > $O/s8.json 2>$O/s8_err.log || { tail -5 $O/s8_err.log; cp /tmp/occ3.so whisperx_mlx_amd/libwxhip.so; exit 1; }
    python -c "
import json;d=json.loads(open('$O/s8.json').read().strip().splitlines()[-1]);m=d['roofline_more']
print('$v','value',d['value'],'ms/step',d['ms_per_step'],'stages',d['stages_ms'],'single-stream encode',d['stages_ms_single_stream']['encode'],'attention probe ms',m['enc_attention']['ms'],'fc1 probe ms',m['enc_fc1_gemm']['ms'])" | tee -a $O/r05_ab_attn_occupancy_in_job.txt
  done
done
cp /tmp/occ3.so whisperx_mlx_amd/libwxhip.so
