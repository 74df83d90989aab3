mkdir -p gpurun_out/r04_full
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r04_full/pytest_gpu.txt
cat gpurun_out/r04_full/pytest_gpu.txt
timeout 900 python bench.py > gpurun_out/r04_full/bench.json 2> gpurun_out/r04_full/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_full/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"], indent=0)[:3000])
PY
tail -3 gpurun_out/r04_full/bench.err
