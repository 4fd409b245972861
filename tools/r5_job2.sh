#!/bin/bash
set -u
OUT=gpurun_out/r5job2
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 600 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_wgrad.py tests/test_gpu_trainer.py tests/test_gpu_graph.py tests/test_gpu_parity.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
$T 300 python3 tools/attn_bwd_stats.py > $OUT/attn_bwd_stats.txt 2>&1
$T 300 python3 tools/g17_bf16_err.py > $OUT/g17_bf16_err.txt 2>&1
$T 300 python3 tools/ctc_fullsize_err.py > $OUT/ctc_fullsize_err.txt 2>&1
$T 300 python3 tools/bench_wgrad_batch.py > $OUT/bench_wgrad_batch.txt 2>&1
cat $OUT/bench_wgrad_batch.txt
for spec in "ASR_AMD_TN_BATCH=0" "ASR_AMD_TN_BATCH=1" "ASR_AMD_TN_BATCH=1 ASR_AMD_TN_BATCH_TILES=80" "ASR_AMD_TN_BATCH=1 ASR_AMD_TN_BATCH_TILES=240" "ASR_AMD_TN_BATCH=1 ASR_AMD_TN_BATCH_WGS=128" "ASR_AMD_TN_BATCH=0" "ASR_AMD_TN_BATCH=1"; do
  echo "== $spec"
  ( export $spec; $T 300 python3 bench.py --brief --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['config']['launch_calibration_ms'], d['losses_last_step'])" )
done > $OUT/tn_batch_ab.txt 2>&1
cat $OUT/tn_batch_ab.txt
