#!/bin/bash
# NOTE: the WX_* environment knobs used here exist in LAB builds only (python tools/build_lab.py env WX_LAB_ENV; then
# run with the lab library: whisperx_mlx_amd._lib.LIB_PATH / tools/ab_lib.py).  The measurements in profiles/r05_ab_*.txt were
# taken while the knobs were still compiled into the round's working library.
# Round 5, GPU session 2: the suite again (attention with the peeled last tile), per-kernel times of a 112-row single pass
# with the one-pass GEMV and with the row-group kernels (rocprofv3), attention / encoder probes, vad_mix with and without.
mkdir -p gpurun_out
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== full GPU suite"; date
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r05_t5.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05_t5.log; tail -4 $O/r05_t5.log
echo "== encoder / attention probes"; date
AB_PIPE_ONLY=1 timeout -k 10 200 python tools/ab_gemm_pipe.py 2>&1 | grep rows | tee $O/r05_probe_attention.txt
echo "== per-kernel: 112 rows x 1, wide GEMV on / off"; date
for env in "" "WX_NO_WIDE_GEMV=1"; do
  tag=$( [ -z "$env" ] && echo wide || echo rowgroups )
  env $env timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o r05 -- python bench.py --steps 7 --warmup 7 --rows-per-pass 112 --streams 1 --no-cpu-baseline --no-extra --no-align > $O/r05_prof_$tag.json 2> $O/r05_prof_$tag.err
  echo "rc=$?"; f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-160
  cp "$f" $O/r05_kernel_stats_112x1_$tag.csv 2>/dev/null
  rm -rf $O/prof_$tag
done
echo "== vad_mix / config4 with and without the one-pass GEMV"; date
for env in "" "WX_NO_WIDE_GEMV=1"; do
  env $env timeout -k 10 300 python bench.py --steps 6 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$env'.ljust(18), 'value', d['value'], 'vad_mix', d['vad_mix']['value'], d['vad_mix']['wall_ms'], 'config4', d['config4']['value'], d['config4']['asr_only_ms'], d['config4']['align_stage_ms'], 'job30', d['job_30min']['resident']['value'], 'b16', d['value_batch16']['value'], 'config2', d['config2']['value'], 'mem', d.get('gpu_memory_gb'))
" | tee -a $O/r05_ab_vadmix.txt
done
date
