#!/bin/bash
# Timing ablation of proj_heads_rows_kernel (ffn.hip): the chunk loop rebuilt with pieces left out (wrong results on purpose) - which
# part of a chunk sets its length.  Needs the library built with -DHEADS_ABLATE (ASR_AMD_EXTRA_HIPCC_FLAGS=-DHEADS_ABLATE python -c
# "import __graft_entry__ as g; g.build()"); bits: 1 no LDS-DMA, 2 no fragment reads, 4 nothing leaves, 8 no MFMA, 16 no global
# stores, 32 no LDS tile round trip.
for a in 0 1 2 4 8 5 6 7 15 16 32 48; do echo "ABL=$a"; ASR_AMD_HEADS_ABL=$a timeout -k 5 60 python tools/bench_heads.py 2>&1 | tail -2; done
