#!/bin/bash
# round 5, session 7 (lab build, -DWX_LAB_ENV): the fused attention launch at ONE block per CU (dynamic LDS padded to 84 KiB)
# against two -- does leaving half the wave slots free shorten the other passes' GEMV chains by more than the launch loses?
set -o pipefail
O=gpurun_out
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-align"
for r in 1 2; do
  for kb in 0 84; do
    WX_XATTN_LDS_KB=$kb timeout -k 10 200 $B > $O/s7.json 2>$O/s7_err.log || { tail -5 $O/s7_err.log; exit 1; }
    python -c "
import json;d=json.loads(open('$O/s7.json').read().strip().splitlines()[-1]);r=d['roofline']
print('WX_XATTN_LDS_KB=$kb'.ljust(20),'value',d['value'],'ms/step',d['ms_per_step'],'launch in flight us',r['avg_launch_us'],'alone us',r['alone']['us'])" | tee -a $O/r05_ab_xattn_one_block_per_cu.txt
  done
done
