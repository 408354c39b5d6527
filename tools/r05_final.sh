#!/bin/bash
# round 5, final state: the whole GPU suite, then the driver's bench command (python bench.py --steps 20 --warmup 5)
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q -rs > $O/r05_final_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/r05_final_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r05_final_bench.json 2> $O/r05_final_bench.err; rc=$?
echo "bench rc=$rc"; python tools/show_bench.py $O/r05_final_bench.json 2>/dev/null | head -30
exit $rc
