mkdir -p gpurun_out/r04_full
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r04_full/pytest_gpu.txt
cat gpurun_out/r04_full/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r04_full/bench.json 2> gpurun_out/r04_full/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_full/bench.json").read().strip().splitlines()[-1])
s=d["summary"]
print(s["checks_per_s"], d["roofline"]["traffic"], d["roofline"].get("valu",{}).get("frac"))
print(json.dumps(s["env_steps_per_s"]))
print(json.dumps(s["rollout_agent_steps_per_s"]), round(d["rollout_async_dyn"]["env_steps_per_s"]))
PY
tail -2 gpurun_out/r04_full/bench.err
