timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/bench15.log 2>&1; tail -2 gpurun_out/bench15.log
TOP=30 timeout 300 python tools/profile_shapes.py > gpurun_out/shapes10.log 2>&1; head -34 gpurun_out/shapes10.log
