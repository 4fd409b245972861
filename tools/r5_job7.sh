#!/bin/bash
set -u
OUT=gpurun_out/r5job7
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 600 python3 -m pytest tests/test_gpu_vocab_ctc.py tests/test_gpu_vocab_lse.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|Error|rc |assert|^E " $OUT/pytest.log | tail -25
$T 200 python3 tools/bench_vocab.py 2>&1 | grep -v amdgpu
