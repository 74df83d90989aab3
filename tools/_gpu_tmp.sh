timeout 900 python bench.py --no-plan --no-rollout 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['env_step'].items():
    if isinstance(v,dict) and 'dynamics_contacts' in v:
        for kk in ('dynamics','dynamics_contacts'):
            c=v[kk]; print(k, kk, round(c['steps_per_s']), [round(x,2) for x in c['ms_per_batch_passes']], c.get('parity_mismatches_vs_oracle'))
"
