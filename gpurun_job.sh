timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stem" 2>&1 | tail -5
for f in "USE_STEM_POOL=1" "USE_STEM_POOL=0" "USE_STEM_POOL=1" "USE_STEM_POOL=0"; do
COBEVT_FLAGS=$f timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
done
