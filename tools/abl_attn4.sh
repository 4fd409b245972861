#!/bin/bash
# Timing-only ablations of the generated attention forward (tools/gen_attn_fwd4.py --abl N): which piece's removal buys time?
#   1 no MFMA   2 no softmax VALU   4 no fragment reads   8 no LDS-DMA   16 no barrier / vmcnt wait
CS=end-to-end_asr_pytorch_amd/csrc
for m in "$@"; do
  python tools/gen_attn_fwd4.py --abl $m /tmp/attn4_abl$m.inc 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DATTN4_INC="\"/tmp/attn4_abl$m.inc\"" -c $CS/attention_fwd4.hip -o /tmp/attn4_abl$m.o 2>/tmp/attn4_abl$m.err || { echo "abl $m: compile failed"; tail -5 /tmp/attn4_abl$m.err; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_abl$m.so /tmp/attn4_abl$m.o $(ls $CS/build/*.o | grep -v "/attention_fwd4.hip.o")
  r=$(ASR_AMD_LIB=/tmp/libasr_abl$m.so timeout 120 python tools/bench_ops.py attn 2>&1 | grep '"attention_fwd"' | head -1 | python -c "import sys,json; print(' '.join(str(json.loads(l)['us']) for l in sys.stdin))")
  echo "ABL=$m  us: $r"
done
