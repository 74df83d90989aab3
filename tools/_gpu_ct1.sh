mkdir -p gpurun_out/r04_ct
timeout 600 python -m pytest tests/test_gpu_dyn.py -x -q -k "contact" 2>&1 | tail -30 > gpurun_out/r04_ct/test.log
cat gpurun_out/r04_ct/test.log
