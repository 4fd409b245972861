#!/bin/bash
# same-box A/B of the built library against another build of the same ABI: tools/ab_lib.sh path/to/other.so "models" [pairs]
OTHER=$1; MODELS=${2:-s1}; PAIRS=${3:-4}
cd $GRAFT_REPO_ROOT
for m in $MODELS; do
for i in $(seq $PAIRS); do
for v in new base; do
  if [ $v = base ]; then export ASR_AMD_LIB=$PWD/$OTHER; else unset ASR_AMD_LIB; fi
  r=$(timeout 300 python bench.py --brief --model $m --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/ab_err_$v.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$m $v ms=$r"
done
done
done
