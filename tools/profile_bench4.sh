#!/bin/bash
# Round 3: rocprofv3 --kernel-trace --stats of the PRODUCT configuration of bench.py (default passes in flight),
# the same command whose `value` is reported, plus the same with 1 / 2 / 3 passes in flight for the gap analysis.
# Output: gpurun_out/prof_r03/ (stats csv and the overlap reports are copied to profiles/).
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for N in ${STREAMS:-4 1 2 3}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench$N" -o bench$N -- python3 "$R/bench.py" --streams $N --steps 8 --warmup 4 --no-cpu-baseline --no-extra --no-align > "$O/bench$N.log" 2>&1
  echo "bench$N trace rc=$?"
  (cd "$R" && python3 tools/trace_overlap.py "$O/bench$N" --inflight $N > "$O/bench${N}_overlap.txt" 2>&1)
  find "$O/bench$N" -name "*_kernel_trace.csv" -delete
done
du -sh "$O"; ls -R "$O" | head -40
