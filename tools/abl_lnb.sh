#!/bin/bash
# add_layernorm_bwd with different persistent-workgroup caps (-DLNB_MAX_BLOCKS=n builds): usage tools/abl_lnb.sh 256 128 64
CS=end-to-end_asr_pytorch_amd/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLNB_MAX_BLOCKS=$n -c $CS/backward.hip -o /tmp/bw_$n.o 2>/dev/null && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_lnb$n.so /tmp/bw_$n.o $(ls $CS/build/*.o | grep -v "/backward.hip.o")
  echo "== LNB_MAX_BLOCKS=$n"; ASR_AMD_LIB=/tmp/libasr_lnb$n.so timeout 120 python3 tools/bench_ln_bwd.py 2>&1 | grep -v amdgpu
done
