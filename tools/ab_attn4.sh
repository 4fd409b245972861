#!/bin/bash
# A/B builds of attention_fwd4.hip (flags per variant), each checked (tools/check_attn4.py) and timed (tools/bench_ops.py attn) in turn.
# usage: tools/ab_attn4.sh "name1:-DFLAG=1 -DX=2" "name2:..."     (run on the GPU box; the rest of the library comes from csrc/build)
CS=end-to-end_asr_pytorch_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $CS/attention_fwd4.hip -o /tmp/attn4_$name.o 2>/tmp/attn4_$name.err || { echo "$name: compile failed"; tail -5 /tmp/attn4_$name.err; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_$name.so /tmp/attn4_$name.o $(ls $CS/build/*.o | grep -v "/attention_fwd4.hip.o")
  echo "== $name [$flags]"
  ASR_AMD_LIB=/tmp/libasr_$name.so timeout 300 python tools/check_attn4.py 2>&1 | tail -16
  ASR_AMD_LIB=/tmp/libasr_$name.so timeout 300 python tools/bench_ops.py attn 2>&1 | grep '"attention_fwd"' | head -2
done
