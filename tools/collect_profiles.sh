#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the bench command, single stream (per-kernel in-situ durations)
#   2. --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes over isolated launches of the hot kernels (tools/probe_kernels.py)
#   3. --pmc TCC_EA0_RDREQ / TCC_EA0_RDREQ_32B (request-size split behind FETCH_SIZE) for the LDS-DMA GEMM and the streaming kernel
#   4. --kernel-trace --stats of the wav2vec2 forward + CTC alignment
# Output: gpurun_out/prof_r02/...   (summaries are copied into profiles/ by hand afterwards)
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r02"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bench" -o bench -- python3 "$R/bench.py" --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-align > "$O/bench.log" 2>&1
echo "bench trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_$C" -o p -- python3 "$R/tools/probe_kernels.py" "fused cq+xattn" "cross-attn split2" "v1 LN+fc1" "v1 fc2 tn8 w16" "v2 logits" "enc " > "$O/pmc_$C.log" 2>&1
  echo "pmc $C rc=$?"
done
rocprofv3 --list-avail > "$O/avail.txt" 2>&1
for C in TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/pmc_$C" -o p -- python3 "$R/tools/probe_kernels.py" "fused cq+xattn" "enc fc2" > "$O/pmc_$C.log" 2>&1
  echo "pmc $C rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/w2v" -o w2v -- python3 "$R/tools/probe_w2v.py" > "$O/w2v.log" 2>&1
echo "w2v trace rc=$?"
# summaries on the box (the raw traces exceed what gpurun copies back)
cd "$R"
python3 tools/make_pmc_summary.py "$O/pmc_summary.json" "b16=16:$O/pmc_FETCH_SIZE:$O/pmc_WRITE_SIZE" > "$O/pmc_summary.log" 2>&1
python3 tools/pmc_requests.py "$O/pmc_TCC_EA0_RDREQ_sum" "$O/pmc_TCC_EA0_RDREQ_32B_sum" "$O/pmc_FETCH_SIZE" > "$O/rdreq_split.json" 2> "$O/rdreq_split.err"
grep -i -E "TCC_EA0_RDREQ|FETCH_SIZE|WRITE_SIZE" "$O/avail.txt" | head -40 > "$O/avail_tcc.txt"
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*counter_collection.csv" -delete
rm -f "$O/avail.txt"
du -sh "$O"; ls -R "$O" | head -40
