#!/bin/bash
# Round 5, GPU session 5: the suite on the final kernels, the driver's bench command, the aligner's cuts once more.
mkdir -p gpurun_out
O=gpurun_out
echo "== full GPU suite"; date
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r05_t6.log 2>&1; echo "pytest rc=$?" | tee -a $O/r05_t6.log; tail -4 $O/r05_t6.log
echo "== bench"; date
timeout -k 10 500 python bench.py --steps 20 > $O/r05_b3.json 2> $O/r05_b3.err; echo "bench rc=$?"; python tools/show_bench.py $O/r05_b3.json | cut -c1-330
python -c "
import json
d=[json.loads(l) for l in open('$O/r05_b3.json') if l.startswith('{')][-1]
print('config2', json.dumps(d.get('config2'))[:600]); print('mem', d.get('gpu_memory_gb')); print('tail', d.get('gather_tail_ms'))"
echo "== aligner cuts"; date
timeout -k 10 300 python tools/ab_align_cuts.py 2>&1 | grep -v "^Failed to align" | tee $O/r05_ab_align_cuts2.txt | head -30
date
