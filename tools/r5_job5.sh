#!/bin/bash
set -u
OUT=gpurun_out/r5job5
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
export ASR_AMD_TN_BATCH=0
$T 300 python3 -m pytest tests/test_gpu_attention_fwd4.py tests/test_gpu_parity.py -m gpu -q -k "attention" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for spec in "X=0" "ASR_AMD_CTC_GRAD_WGS_PER_CU=6" "ASR_AMD_CTC_GRAD_WGS_PER_CU=4" "ASR_AMD_GEMM_WGS=256" "ASR_AMD_GEMM_WGS=256 ASR_AMD_CTC_GRAD_WGS_PER_CU=6" "ASR_AMD_GEMM_WGS=256 ASR_AMD_CTC_GRAD_WGS_PER_CU=4" "ASR_AMD_GEMM_WGS=384 ASR_AMD_CTC_GRAD_WGS_PER_CU=5" "X=0"; do echo "== $spec"; ( export $spec; $T 200 python3 tools/step_segments.py 2>&1 | grep -v amdgpu ); done > $OUT/segments.txt 2>&1
cat $OUT/segments.txt
