timeout 1200 python -m pytest tests/test_gpu_rollout.py -x -q 2>&1 | tail -3
