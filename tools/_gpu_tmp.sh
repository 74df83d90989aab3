R=$PWD
mkdir -p gpurun_out/r04_flip
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
rm -rf /tmp/flip_$i
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/flip_$i -o flip -- python $R/bench.py --no-rollout > /tmp/flip_$i.json 2>/tmp/flip_$i.err
python - <<PY
import json,csv,glob
d=json.loads(open("/tmp/flip_$i.json").read().strip().splitlines()[-1])
print("run $i", {k: round(v['dynamics_contacts']['gpu_ms_per_batch'],2) for k,v in d['env_step'].items() if isinstance(v,dict) and 'dynamics_contacts' in v})
f=glob.glob("/tmp/flip_$i/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_env_dyn_ct" in r["Name"] or "k_env_step<1>" in r["Name"]:
        print("   ", r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["MinNs"], r["MaxNs"])
PY
done
