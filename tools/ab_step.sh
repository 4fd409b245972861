#!/bin/bash
# A/B of the S1 training step under environment switches: tools/ab_step.sh "NAME=VAL ..." "NAME=VAL ..." ...  ("-" = defaults)
for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    r=$(env $e timeout -k 5 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('eager_ms_per_step'), d.get('replay_ms_per_step'))")
    echo "[$cfg] ms_per_step eager replay: $r"
done
