timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench20.log 2>&1; tail -1 gpurun_out/bench20.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
