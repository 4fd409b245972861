#!/bin/bash
# Regenerates the judged artifacts on a GPU box into gpurun_out/final/ (copy what should be kept into profiles/rN/ afterwards).
# usage (from the repo root on the GPU box): bash tools/refresh_profiles.sh
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2> $O/bench_fwd.err
python tools/bench_ops.py > $O/bench_ops.jsonl 2> $O/bench_ops.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/stats.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/write.log 2>&1
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$F" "$W" $O/pmc_traffic_train_s1.json > $O/pmc_summary.log 2>&1
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/bench_train_kernel_stats.csv
# the raw traces are large: keep the summaries only
rm -rf $O/fetch $O/write $O/stats
ls -la $O
