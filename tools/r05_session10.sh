#!/bin/bash
set -o pipefail
timeout -k 10 300 python tools/lab_gemm_timeline.py 16 > gpurun_out/r05_lab_gemm_timeline.txt 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r05_lab_gemm_timeline.txt | tail -60; exit $rc
