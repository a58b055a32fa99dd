timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chain" 2>&1 | tail -2
for f in "ROW_CHAIN_ROWS=0" "ROW_CHAIN_ROWS=64"; do
COBEVT_FLAGS=$f timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
done
