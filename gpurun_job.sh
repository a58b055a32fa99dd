timeout 900 python bench.py > gpurun_out/bench21.log 2>&1; tail -1 gpurun_out/bench21.log
