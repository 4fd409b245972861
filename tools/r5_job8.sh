#!/bin/bash
set -u
OUT=gpurun_out/r5job8
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 1200 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc |^E  " $OUT/pytest.log | tail -15
$T 200 python3 tools/bench_vocab.py 2>&1 | grep -v amdgpu
$T 300 python3 bench.py --brief --steps 40 --warmup 5 2>>$OUT/bench.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('S1', d['ms_per_step'], d['losses_last_step'], d['config']['launch_calibration_ms'])"
cd /tmp; $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt_aishell -- python3 $GRAFT_REPO_ROOT/bench.py --model aishell --brief --steps 20 --warmup 3 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
KS=$(find $OUT/kt_aishell -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && head -30 $KS | cut -d, -f1-4 | cut -c1-150 > $OUT/aishell_kernel_stats.txt; rm -rf $OUT/kt_aishell
cat $OUT/aishell_kernel_stats.txt
