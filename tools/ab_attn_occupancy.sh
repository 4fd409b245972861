#!/bin/bash
# A/B: the v2 attention forward at 2 / 3 / 4 resident workgroups per CU (launch bounds -> register budget), against v3
CS=end-to-end_asr_pytorch_amd/csrc
for m in 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DATTN_V2_MINB=$m -c $CS/attention.hip -o /tmp/attn_m$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_m$m.so /tmp/attn_m$m.o $(ls $CS/build/*.o | grep -v "/attention.hip.o")
  echo "MINB=$m (v2)"; ASR_AMD_LIB=/tmp/libasr_m$m.so ASR_AMD_ATTN_V3=0 timeout 120 python tools/bench_ops.py attn 2>&1 | grep attention_fwd | head -2
done
echo "v3"; timeout 120 python tools/bench_ops.py attn 2>&1 | grep attention_fwd | head -2
