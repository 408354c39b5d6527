#!/bin/bash
# round 5: SQ counters of the encoder kernels as they are now (gemm_pipe_kernel, attn_full_kernel at three blocks per CU);
# one rocprofv3 --pmc pass (kernel trace only), summarised by tools/sq_summary.py -> profiles/r05_sq_encoder_kernels.json
set -o pipefail
R=$(pwd); O=$R/gpurun_out/r05_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/pmc -o p -- python3 $R/tools/probe_kernels.py "enc " > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
tail -4 $O/pmc.log
cd $R && python tools/sq_summary.py $O/pmc gpurun_out/r05_sq_encoder_kernels.json
find $O -name "*.csv" -size +2M -delete
