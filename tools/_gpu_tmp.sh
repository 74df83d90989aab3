mkdir -p gpurun_out/r04_full
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04_full/pytest_gpu.txt
cat gpurun_out/r04_full/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r04_full/bench.json 2> gpurun_out/r04_full/bench.err
tail -c 1500 gpurun_out/r04_full/bench.json
tail -5 gpurun_out/r04_full/bench.err
