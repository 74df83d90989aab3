mkdir -p gpurun_out/r04_sweep
timeout 2400 python tools/ct_parity_sweep.py 1024 4 2>&1 | grep -v amdgpu | tee gpurun_out/r04_sweep/ct_parity_sweep.txt
