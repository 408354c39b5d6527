#!/bin/bash
# Round 3 rocprofv3 evidence (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the DRIVER's bench command (default scheduler: 5 passes of 64 rows, three in flight)
#      -> per-kernel in-situ durations of the product configuration + tools/trace_overlap.py report
#   2. the same with --streams 1 --rows-per-pass 16 (one 16-row pass alone: round 2's profile, for comparison)
#   3. --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes over isolated launches of the hot kernels at 64 and 16 rows
# Output: gpurun_out/prof_r03/  (the summaries are copied into profiles/ afterwards)
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-align > "$O/bench.log" 2>&1
echo "bench trace rc=$?"
(cd "$R" && python3 tools/trace_overlap.py "$O/bench" --inflight 3 > "$O/bench_overlap.txt" 2>&1)
# the plan of the traced run (bench.py only quotes the in-flight window for a run of the same plan)
python3 - "$O/bench.log" "$O/bench_plan.json" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
r = d["config"]["rows_per_pass"]
json.dump({"rows": r if isinstance(r, list) else [r] * int(round(d["steps"] * 16 / r)), "passes_in_flight": d["config"]["passes_in_flight_per_gpu"],
           "value_under_the_profiler": d["value"]}, open(sys.argv[2], "w"))
PY
find "$O/bench" -name "*_kernel_trace.csv" -delete
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench1" -o bench1 -- python3 "$R/bench.py" --streams 1 --rows-per-pass 16 --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-align > "$O/bench1.log" 2>&1
echo "bench1 trace rc=$?"
find "$O/bench1" -name "*_kernel_trace.csv" -delete
for B in 128 64 16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    PROBE_B=$B rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_${C}_b$B" -o p -- python3 "$R/tools/probe_kernels.py" "fused cq+xattn" "cross-attn split2" "v1 LN+fc1" "v1 fc2 tn8 w16" "v2 logits" > "$O/pmc_${C}_b$B.log" 2>&1
    echo "pmc $C b$B rc=$?"
  done
done
cd "$R"
python3 tools/make_pmc_summary.py "$O/pmc_summary.json" "b128=128:$O/pmc_FETCH_SIZE_b128:$O/pmc_WRITE_SIZE_b128" "b64=64:$O/pmc_FETCH_SIZE_b64:$O/pmc_WRITE_SIZE_b64" "b16=16:$O/pmc_FETCH_SIZE_b16:$O/pmc_WRITE_SIZE_b16" > "$O/pmc_summary.log" 2>&1
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*counter_collection.csv" -delete
du -sh "$O"; ls "$O"
