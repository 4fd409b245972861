#!/bin/bash
set -u
OUT=gpurun_out/r5job11
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_vocab_ctc.py tests/test_gpu_edge_cases.py -m gpu -q -k "ctc or CTC" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc " $OUT/pytest.log | tail -5
for v in 0 1 0 1; do echo "== ASR_AMD_CTC_DBG=$v"; ASR_AMD_CTC_DBG=$v $T 120 python3 tools/ab_ctc.py 2>&1 | grep -v amdgpu; done
bash tools/kt_ctc.sh default ASR_AMD_CTC_DBG=1
