#!/bin/bash
# Timing variants of the generated attention forward: each spec is "ABL,DMA_AT,NW[,check]" (tools/gen_attn_fwd4.py --abl / --dma; NW by
# ASR_AMD_ATTN_NW).  ABL != 0 builds are timing-only (results garbage): 1 no MFMA  2 no softmax VALU  4 no fragment reads  8 no LDS-DMA
# 16 no barrier / vmcnt wait.  "check" also runs tools/check_attn4.py on that build.
CS=end-to-end_asr_pytorch_amd/csrc
for spec in "$@"; do
  IFS=, read abl dma nw chk <<< "$spec"
  tag=${abl}_${dma}
  if [ ! -f /tmp/libasr_v$tag.so ]; then
    python tools/gen_attn_fwd4.py --abl $abl --dma $dma --out /tmp/attn4_v$tag.inc 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DATTN4_INC="\"/tmp/attn4_v$tag.inc\"" -c $CS/attention_fwd4.hip -o /tmp/attn4_v$tag.o 2>/tmp/attn4_v$tag.err || { echo "$spec: compile failed"; grep -m3 error /tmp/attn4_v$tag.err; continue; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_v$tag.so /tmp/attn4_v$tag.o $(ls $CS/build/*.o | grep -v "/attention_fwd4.hip.o")
  fi
  if [ "$chk" = "check" ]; then ASR_AMD_ATTN_NW=$nw ASR_AMD_LIB=/tmp/libasr_v$tag.so timeout 200 python tools/check_attn4.py 2>&1 | grep -c " ok$" | sed "s/^/   check_attn4 ok lines (of 14): /"; fi
  r=$(ASR_AMD_ATTN_NW=$nw ASR_AMD_LIB=/tmp/libasr_v$tag.so timeout 120 python tools/bench_ops.py attn 2>&1 | grep '"attention_fwd"' | python -c "import sys,json; print(' '.join(str(json.loads(l)['us']) for l in sys.stdin if 'dropout\": 0.0' in l))")
  echo "abl=$abl dma=$dma nw=$nw  us [1000x1000, 250x250, 51x1000, 51x51c]: $r"
done
