timeout 600 python -m pytest tests/test_gpu_dgrad_rows.py tests/test_gpu_ffn_fused.py -q 2>&1 | tail -n 6
for i in 1 2 3; do for v in 1 0; do
ASR_AMD_LN_FROM_Y=$v python bench.py --steps 30 --no-cpu-baseline --no-also --brief 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FROM_Y=$v ms_per_step', j['ms_per_step'])"
done; done
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep "passed\|failed"
