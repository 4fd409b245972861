#!/bin/bash
# Ablation of the attention forward loops (timing only, results are garbage): which piece's removal buys time?
#   1 no score MFMAs   2 no PV MFMAs   4 no exp (v2)   8 no LDS fragment reads (v2)   16 no LDS-DMA (v2)   32 no per-tile barrier (v2)
# usage: tools/ablate_attn.sh [v3]      (v3: only bits 1 and 2 exist there)
CS=end-to-end_asr_pytorch_amd/csrc
V3=0; MASKS="0 1 2 3 4 8 16 32 48 56 7 63"
if [ "${1:-}" = "v3" ]; then V3=1; MASKS="0 3 8 16 32 56 59"; fi
for m in $MASKS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DATTN_ABL=$m -c $CS/attention.hip -o /tmp/attn_a$m.o 2>/dev/null || { echo "ABL=$m: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_a$m.so /tmp/attn_a$m.o $(ls $CS/build/*.o | grep -v "/attention.hip.o")
  r=$(ASR_AMD_LIB=/tmp/libasr_a$m.so ASR_AMD_ATTN_V3=$V3 timeout 120 python tools/bench_ops.py attn 2>&1 | grep attention_fwd | head -2 | python -c "import sys,json; print(' '.join(str(json.loads(l)['us']) for l in sys.stdin))")
  echo "v3=$V3 ABL=$m  us (no dropout, dropout): $r"
done
