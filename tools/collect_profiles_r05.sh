#!/bin/bash
# Round 5 rocprofv3 evidence (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the DRIVER's bench command (value: 112 + 112 + 96 rows x 3 in flight; value_batch16: 16 rows
#      x 4 in flight; job_30min, vad_mix) -> per-kernel stats, tools/trace_overlap.py report of the three-in-flight window,
#      and tools/trace_by_width.py: ONE ROW PER KERNEL AND LAUNCH WIDTH (whole trace and in-flight window side by side)
#   2. --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes over isolated launches of the hot kernels at 128 / 112 / 16 rows
# Output: gpurun_out/prof_r05/  (the summaries are copied into profiles/ afterwards)
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r05"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# (a) the `value` phase alone (--no-extra): per-kernel stats + what three wide passes in flight do to each other
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -o bench -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-align > "$O/bench.log" 2>&1
echo "bench trace rc=$?"
(cd "$R" && python3 tools/trace_overlap.py "$O/bench" --inflight 3 > "$O/bench_overlap.txt" 2>&1)
# (b) the whole driver command (value, value_batch16, job_30min, vad_mix): one row per kernel and launch width
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench_all" -o bench_all -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-align > "$O/bench_all.log" 2>&1
echo "bench_all trace rc=$?"
(cd "$R" && python3 tools/trace_by_width.py "$O/bench_all" "$O/bench_kernel_stats_by_width.csv" --inflight 3 > "$O/by_width.log" 2>&1)
find "$O/bench_all" -name "*_kernel_trace.csv" -delete
[ -n "${SKIP_PMC:-}" ] && { find "$O" -name "*_kernel_trace.csv" -delete; du -sh "$O"; exit 0; }
# the plan of the traced run (bench.py only quotes the in-flight window for a run of the same plan)
python3 - "$O/bench.log" "$O/bench_plan.json" <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
r = d["config"]["rows_per_pass"]
json.dump({"rows": r if isinstance(r, list) else [r] * int(round(d["steps"] * 16 / r)), "passes_in_flight": d["config"]["passes_in_flight_per_gpu"],
           "value_under_the_profiler": d["value"], "value_batch16_under_the_profiler": d.get("value_batch16", {}).get("value")}, open(sys.argv[2], "w"))
PY
find "$O/bench" -name "*_kernel_trace.csv" -delete
for B in 128 112 16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    PROBE_B=$B rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_${C}_b$B" -o p -- python3 "$R/tools/probe_kernels.py" "fused cq+xattn" "cross-attn split2" "v1 LN+fc1" "v1 fc2 tn8 w16" "v2 logits" > "$O/pmc_${C}_b$B.log" 2>&1
    echo "pmc $C b$B rc=$?"
  done
done
cd "$R"
python3 tools/make_pmc_summary.py "$O/pmc_summary.json" "b128=128:$O/pmc_FETCH_SIZE_b128:$O/pmc_WRITE_SIZE_b128" "b112=112:$O/pmc_FETCH_SIZE_b112:$O/pmc_WRITE_SIZE_b112" "b16=16:$O/pmc_FETCH_SIZE_b16:$O/pmc_WRITE_SIZE_b16" > "$O/pmc_summary.log" 2>&1
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*counter_collection.csv" -delete
du -sh "$O"; ls "$O"
