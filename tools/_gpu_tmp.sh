mkdir -p gpurun_out/r04_full
timeout 900 python bench.py > gpurun_out/r04_full/bench.json 2> gpurun_out/r04_full/bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_full/bench.json").read().strip().splitlines()[-1])
s=d["summary"]
print(s["checks_per_s"], d["ms_per_step"], d["roofline"]["traffic"] is not None)
print(json.dumps({k:{kk:round(vv) for kk,vv in v.items()} for k,v in s["env_steps_per_s"].items()}))
print(json.dumps({k:round(v) for k,v in s["rollout_agent_steps_per_s"].items()}))
PY
tail -2 gpurun_out/r04_full/bench.err
