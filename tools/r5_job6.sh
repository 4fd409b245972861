#!/bin/bash
set -u
OUT=gpurun_out/r5job6
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 600 python3 -m pytest tests/test_gpu_vocab_ctc.py tests/test_gpu_vocab_lse.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|Error|rc |assert" $OUT/pytest.log | tail -15
$T 600 python3 -m pytest tests/test_gpu_trainer.py tests/test_gpu_graph.py tests/test_gpu_dropout.py tests/test_gpu_fullsize.py tests/test_gpu_lds_poison.py -m gpu -q > $OUT/pytest2.log 2>&1; echo "pytest rc $?" >> $OUT/pytest2.log
grep -E "passed|failed|FAILED|rc " $OUT/pytest2.log | tail -8
for i in 1 2; do $T 300 python3 bench.py --brief --steps 40 --warmup 5 2>>$OUT/bench.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('S1', d['ms_per_step'], d['losses_last_step'], d['config']['launch_calibration_ms'])"; done
$T 200 python3 tools/step_segments.py 2>&1 | grep -v amdgpu
SIDE_INLINE=1 $T 200 python3 tools/step_segments.py 2>&1 | grep -v amdgpu
