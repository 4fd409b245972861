#!/bin/bash
# timing ablation of proj_heads_rows_kernel (wrong results on purpose): which part of a chunk sets its length
for a in 0 1 2 4 8 3 5 6 7 14 15; do echo "ABL=$a"; ASR_AMD_HEADS_ABL=$a timeout -k 5 60 python tools/bench_heads.py 2>&1 | tail -2; done
