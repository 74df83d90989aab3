mkdir -p gpurun_out/r04_sweep
timeout 1200 python tools/parity_sweep.py 2>&1 | grep -v amdgpu | tail -20 > gpurun_out/r04_sweep/parity_sweep.txt; tail -4 gpurun_out/r04_sweep/parity_sweep.txt
timeout 900 python tools/plan_parity_sweep.py 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r04_sweep/plan_parity_sweep.txt; tail -3 gpurun_out/r04_sweep/plan_parity_sweep.txt
timeout 900 python tools/motion_parity_sweep.py 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/r04_sweep/motion_parity_sweep.txt; tail -3 gpurun_out/r04_sweep/motion_parity_sweep.txt
