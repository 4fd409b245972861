#!/bin/bash
# tools/ab_model.sh MODEL "ENV ..." "ENV ..." : brief bench of one model under environment switches ("-" = defaults)
m=$1; shift
for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    r=$(env $e timeout -k 5 300 python bench.py --model $m --brief --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['launch_calibration_ms'])")
    echo "[$m | $cfg] $r"
done
