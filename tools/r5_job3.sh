#!/bin/bash
set -u
OUT=gpurun_out/r5job3
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
$T 600 python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_wgrad.py tests/test_gpu_fullsize.py tests/test_gpu_transformer_plain.py tests/test_gpu_dp.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for spec in "ASR_AMD_TN_BATCH=0" "ASR_AMD_TN_BATCH_TILES=80 ASR_AMD_TN_BATCH_WGS=256" "ASR_AMD_TN_BATCH_TILES=80 ASR_AMD_TN_BATCH_WGS=160" "ASR_AMD_TN_BATCH_TILES=80 ASR_AMD_TN_BATCH_WGS=128" "ASR_AMD_TN_BATCH_TILES=160 ASR_AMD_TN_BATCH_WGS=320" "ASR_AMD_TN_BATCH_TILES=160 ASR_AMD_TN_BATCH_WGS=480" "ASR_AMD_TN_BATCH_TILES=160 ASR_AMD_TN_BATCH_WGS=256" "ASR_AMD_TN_BATCH=0" "ASR_AMD_TN_BATCH_TILES=80 ASR_AMD_TN_BATCH_WGS=256"; do
  echo "== $spec"
  ( export $spec; $T 300 python3 bench.py --brief --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']['launch_calibration_ms']; print(d['ms_per_step'], 'eager', c['eager_ms'], 'graph', c['graph_ms'], d['losses_last_step'])" )
done > $OUT/tn_batch_ab.txt 2>&1
cat $OUT/tn_batch_ab.txt
