mkdir -p gpurun_out/r04_launches
MOPA_BENCH_ROLLOUT=fused=0 timeout 600 python tools/count_launches.py 100 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" > gpurun_out/r04_launches/torch_form.txt
timeout 600 python tools/count_launches.py 100 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" > gpurun_out/r04_launches/fused_form.txt
head -3 gpurun_out/r04_launches/torch_form.txt | cut -c1-300; head -70 gpurun_out/r04_launches/fused_form.txt | cut -c1-140
