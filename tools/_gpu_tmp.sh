mkdir -p gpurun_out/r04_launches
timeout 1500 python -m pytest tests/test_gpu_env.py tests/test_gpu_rollout.py -x -q 2>&1 | tail -5
timeout 600 python tools/count_launches.py 20 > gpurun_out/r04_launches/after3.txt 2>&1
grep -E "device launches|k_env_exec|k_pb_gen|k_is_valid" gpurun_out/r04_launches/after3.txt | cut -c1-120
