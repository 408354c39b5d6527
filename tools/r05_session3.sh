#!/bin/bash
# NOTE: the WX_* environment knobs used here exist in LAB builds only (python tools/build_lab.py env WX_LAB_ENV; then
# run with the lab library: whisperx_mlx_amd._lib.LIB_PATH / tools/ab_lib.py).  The measurements in profiles/r05_ab_*.txt were
# taken while the knobs were still compiled into the round's working library.
# Round 5, GPU session 3: per-kernel times of a 112-row single pass with the one-pass GEMV and with the row-group kernels
# (rocprofv3 kernel trace), then the poll-window sweep of the fused launch and the GEMM lab knobs (tools/r05_sweeps.sh).
mkdir -p gpurun_out
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== per-kernel: 112 rows x 1, wide GEMV on / off"; date
for env in "" "WX_NO_WIDE_GEMV=1"; do
  tag=$( [ -z "$env" ] && echo wide || echo rowgroups )
  env $env timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o r05 -- python3 bench.py --steps 7 --warmup 7 --rows-per-pass 112 --streams 1 --no-cpu-baseline --no-extra --no-align > $O/r05_prof_$tag.json 2> $O/r05_prof_$tag.err
  echo "rc=$?"; f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); echo "$f"
  [ -n "$f" ] && cp "$f" $O/r05_kernel_stats_112x1_$tag.csv && head -22 "$f" | cut -c1-170
  rm -rf $O/prof_$tag
done
bash tools/r05_sweeps.sh
