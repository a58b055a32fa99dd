for f in "USE_EMBED_GEMM=1" "USE_EMBED_GEMM=0"; do
COBEVT_FLAGS=$f timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
done
