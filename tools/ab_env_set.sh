#!/bin/bash
# same-box A/B of an environment variable being SET (to 1) or unset: tools/ab_env_set.sh VAR "models" [pairs]
VAR=$1; MODELS=${2:-s1}; PAIRS=${3:-3}
cd $GRAFT_REPO_ROOT
for m in $MODELS; do
for i in $(seq $PAIRS); do
for v in unset set; do
  if [ $v = set ]; then export $VAR=1; else unset $VAR; fi
  r=$(timeout 300 python bench.py --brief --model $m --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/ab_err_$v.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$m $VAR $v ms=$r"
done
done
done
