#!/bin/bash
# Timing-only builds of the fused feed-forward forward (csrc/ffn.hip, -DFFN_ABL=n: 1 no fragment LDS reads, 2 no MFMAs, 4 no LDS-DMA;
# results garbage): what the loop's time is made of.  usage: tools/abl_ffn.sh 0 1 2 3 4 5 6 7
CS=end-to-end_asr_pytorch_amd/csrc
for abl in "$@"; do
  if [ ! -f /tmp/libasr_ffn$abl.so ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFFN_ABL=$abl -c $CS/ffn.hip -o /tmp/ffn_abl$abl.o 2>/tmp/ffn_abl$abl.err || { echo "abl=$abl: compile failed"; grep -m3 error /tmp/ffn_abl$abl.err; continue; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libasr_ffn$abl.so /tmp/ffn_abl$abl.o $(ls $CS/build/*.o | grep -v "/ffn.hip.o")
  fi
  r=$(ASR_AMD_LIB=/tmp/libasr_ffn$abl.so timeout 120 python3 tools/bench_ffn_time.py 2>&1 | tail -1)
  echo "FFN_ABL=$abl  $r"
done
