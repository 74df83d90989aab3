timeout 1800 python -m pytest tests/test_gpu_dyn.py -x -q 2>&1 | tail -4
timeout 600 python tools/ct_bench.py 4096 10 8 2>&1 | grep -v amdgpu | grep "env.step" | cut -c1-200
