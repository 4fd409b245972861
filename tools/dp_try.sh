#!/bin/bash
# one 2-rank (gloo, one GPU) bench step under the given env switches; prints the last line or the assertion
for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    r=$(env $e ASR_AMD_DIST_BACKEND=gloo ASR_AMD_DEVICE=0 timeout -k 5 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "AssertionError|\"metric\"" | head -2 | cut -c1-160)
    echo "[$cfg] $r"
done
