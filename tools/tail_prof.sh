#!/bin/bash
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_tail -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --brief > /dev/null 2>&1
cd $R
KT=$(find gpurun_out/kt_tail -name "*kernel_trace.csv" | head -1)
python3 tools/step_tail.py $KT 34 > gpurun_out/step_tail.txt 2>&1; python3 tools/step_streams.py $KT > gpurun_out/step_streams.txt 2>&1
rm -rf gpurun_out/kt_tail
cat gpurun_out/step_tail.txt
