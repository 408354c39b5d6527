#!/bin/bash
# kernel traces of coalesced configurations (rows per pass x passes in flight) + overlap reports
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# CONFIGS="rows,passes ..." e.g. CONFIGS="64,3 16,4"
for C in ${CONFIGS:-64,1 64,2 64,3}; do
  set -- ${C//,/ }
  T="rows$1x$2"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$T" -o $T -- python3 "$R/tools/prof_rows_inflight.py" $1 $2 24 > "$O/$T.log" 2>&1
  echo "$T rc=$? $(grep rows "$O/$T.log" | tail -1)"
  (cd "$R" && python3 tools/trace_overlap.py "$O/$T" --inflight $2 > "$O/${T}_overlap.txt" 2>&1)
  find "$O/$T" -name "*_kernel_trace.csv" -delete
done
