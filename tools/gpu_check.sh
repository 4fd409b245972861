#!/bin/bash
# full GPU test suite + the default bench line (a correctness / regression check after a refactor)
set -u
OUT=gpurun_out/check
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 1200 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc " $OUT/pytest.log | tail -12
$T 300 python3 bench.py --brief --steps 40 --warmup 5 2>$OUT/bench.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('S1', d['ms_per_step'], d['config']['launch_calibration_ms'])"
$T 200 python3 tools/odd_shapes.py 2>&1 | tail -5
