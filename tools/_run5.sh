for i in 1 2 3 4; do for side in 1 0; do
ASR_AMD_DEC_MASK_SIDE=$side python bench.py --steps 30 --no-cpu-baseline --no-also --brief 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SIDE=$side ms_per_step', j['ms_per_step'])"
done; done
