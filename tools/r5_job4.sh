#!/bin/bash
set -u
OUT=gpurun_out/r5job4
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
export ASR_AMD_TN_BATCH=0
for spec in "X=0" "SIDE_INLINE=1" "ASR_AMD_WGRAD_STREAM=0" "X=0" "SIDE_INLINE=1"; do echo "== $spec"; ( export $spec; $T 200 python3 tools/step_segments.py 2>&1 | grep -v amdgpu ); done > $OUT/segments.txt 2>&1
cat $OUT/segments.txt
for spec in "X=0" "ASR_AMD_WGRAD_SIDE_WGS=96 ASR_AMD_TN_GROUP_WGS=96" "ASR_AMD_WGRAD_SIDE_WGS=160 ASR_AMD_TN_GROUP_WGS=160" "ASR_AMD_WGRAD_SIDE_WGS=192 ASR_AMD_TN_GROUP_WGS=192" "ASR_AMD_WGRAD_SIDE_WGS=128 ASR_AMD_TN_GROUP_WGS=256" "ASR_AMD_WGRAD_SIDE_WGS=128 ASR_AMD_TN_GROUP_WGS=64" "X=0"; do
  echo "== $spec"
  ( export $spec; $T 300 python3 bench.py --brief --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']['launch_calibration_ms']; print(d['ms_per_step'], 'eager', c['eager_ms'], 'graph', c['graph_ms'])" )
done > $OUT/side_wgs.txt 2>&1
cat $OUT/side_wgs.txt
