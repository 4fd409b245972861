#!/bin/bash
# Timing-only ablations of the generated attention backward (tools/gen_attn_bwd.py --abl N): kernel-trace durations per variant.
#   1 no MFMA   2 no vector work   4 no fragment reads   8 no LDS-DMA   16 no barrier
CS=end-to-end_asr_pytorch_amd/csrc
for m in "$@"; do
  python tools/gen_attn_bwd.py --abl $m --out /tmp/attn_bwd_abl$m.inc 2>/dev/null
  sed "s#\"attention_bwd_asm.inc\"#\"/tmp/attn_bwd_abl$m.inc\"#" $CS/attention_bwd4.hip > /tmp/attention_bwd4_abl$m.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$CS -c /tmp/attention_bwd4_abl$m.hip -o /tmp/attn_bwd_abl$m.o 2>/tmp/attn_bwd_abl$m.err || { echo "abl $m: compile failed"; grep -m3 error /tmp/attn_bwd_abl$m.err; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_babl$m.so /tmp/attn_bwd_abl$m.o $(ls $CS/build/*.o | grep -v "/attention_bwd4.hip.o")
  echo "== ABL=$m"
  ASR_AMD_LIB=/tmp/libasr_babl$m.so tools/kt_attn.sh 2>&1 | grep "bwd_dkv\|bwd_dq_v4"
  ASR_AMD_LIB=/tmp/libasr_babl$m.so tools/kt_attn.sh --drop 2>&1 | grep "bwd_dkv\|bwd_dq_v4"
done
