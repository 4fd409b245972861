#!/bin/bash
# same-box comparison of several values of one environment variable: tools/ab_vals.sh VAR "v0 v1 ..." [rounds] [model]
VAR=$1; VALS=$2; ROUNDS=${3:-3}; M=${4:-s1}
cd $GRAFT_REPO_ROOT
for i in $(seq $ROUNDS); do
for v in $VALS; do
  r=$(env $VAR=$v timeout 300 python bench.py --brief --model $M --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/ab_err_$v.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$VAR=$v ms=$r"
done
done
