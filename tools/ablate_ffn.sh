#!/bin/bash
# Timing ablation of ffn_fwd_kernel's loop (ffn.hip: FFN_ABL): the library is rebuilt on the GPU box once per variant (wrong results on
# purpose) and tools/bench_ffn.py's forward timings printed.  Bits: 1 no LDS-DMA, 2 no fragment reads, 4 no ReLU / mask / hidden stores,
# 8 no first-product MFMAs, 16 no second-product MFMAs.
for a in ${ABLS:-0 4 1 2 24 7 31}; do
  touch end-to-end_asr_pytorch_amd/csrc/ffn.hip
  ASR_AMD_EXTRA_HIPCC_FLAGS=-DFFN_ABL=$a python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
  echo "FFN_ABL=$a"; timeout -k 5 120 python3 tools/bench_ffn.py 2>&1 | grep "fused forward"
done
touch end-to-end_asr_pytorch_amd/csrc/ffn.hip; python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
