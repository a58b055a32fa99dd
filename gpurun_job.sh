timeout 300 python tools/gemm_probe.py 2>&1 | grep -v amdgpu.ids
