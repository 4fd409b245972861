#!/bin/bash
set -u
OUT=gpurun_out/r5job10
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 600 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc " $OUT/pytest.log | tail -5
for i in 1 2; do $T 300 python3 bench.py --model aishell --brief --steps 40 --warmup 5 2>>$OUT/bench.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('aishell', d['ms_per_step'], d['config']['launch_calibration_ms'])"; done
