"""Where K1's narrow-phase entries come from (CPU, test infrastructure): candidate pairs per state that pass a bounding-sphere
cull / the kernel's sphere + static-AABB cull / an AABB cull with the owner's true extents, and how many really violate the
threshold; then the moving-moving pairs by survival frequency.  `python tools/k1_workload_stats.py [env]`."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, collections
from conftest import sample_states
from mopa_rl_amd.scene import planner_inputs
from oracle import oracle as O
env = [a for a in sys.argv[1:] if not a.startswith("--")][0] if [a for a in sys.argv[1:] if not a.startswith("--")] else "SawyerPushObstacle-v0"
pi = planner_inputs(env); m = pi.model
orc = O.OracleScene(m, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
N=4000
qu,row = sample_states(pi,N//2,1,"uniform"); qn,_=sample_states(pi,N//2,2,"near")
qa=np.concatenate([qu,qn]); thr=pi.spec.contact_threshold
ign=set(tuple(p) for p in pi.ignored_contacts)
never=set((int(a),int(b)) for a,b in m.meta.get('never_violating_pairs',[])) if '--pruned' in sys.argv else set()
pairs=[(int(a),int(b)) for a,b in m.pair_geom if (min(int(m.geom_mjid[a]),int(m.geom_mjid[b])),max(int(m.geom_mjid[a]),int(m.geom_mjid[b]))) not in ign and (int(a),int(b)) not in never]
T=m.geom_type; S=m.geom_size
# static geoms: body has no joint on path to world
def moving(b):
    while b>0:
        if m.body_jntnum[b]>0: return True
        b=int(m.body_parent[b])
    return False
mov=np.array([moving(int(b)) for b in m.geom_body])
def rbound(g):
    t=int(T[g]); s=S[g]
    return {2:s[0],3:s[0]+s[1],5:np.hypot(s[0],s[1]),6:np.linalg.norm(s),0:1e9,7:np.linalg.norm(s)}[t]
def ext(g,R):
    t=int(T[g]); s=S[g]; a=R[:,2]
    if t==2: return np.full(3,s[0])
    if t==3: return s[0]+s[1]*np.abs(a)
    if t==5: return s[1]*np.abs(a)+s[0]*np.sqrt(np.maximum(0,1-a*a))
    if t==6: return np.abs(R)@s
    return np.full(3,rbound(g))
rb=np.array([rbound(g) for g in range(len(T))])
names={0:'pl',2:'sp',3:'cap',5:'cyl',6:'box',7:'mesh'}
cnt=collections.defaultdict(lambda: np.zeros(4))
idx={ (int(a),int(b)):k for k,(a,b) in enumerate(m.pair_geom)}
tot=np.zeros(4)
for i in range(N):
    q=row[0].copy(); q[pi.ref_joint_pos_indexes]=qa[i]
    gp,gm=orc.fk(q); pd=orc.pair_dist(q)
    E=[ext(g,gm[g].reshape(3,3)) for g in range(len(T))]
    for (a,b) in pairs:
        if T[a]==0 or T[b]==0: continue
        d=gp[a]-gp[b]; c=np.linalg.norm(d)
        s0 = c <= rb[a]+rb[b]
        k=names[int(T[a])]+'-'+names[int(T[b])]+('' if (mov[a] and mov[b]) else '/st')
        if not s0: continue
        real = pd[idx[(a,b)]]<=thr
        if mov[a] and mov[b]:
            s1=s2=True
        else:
            st,mv=(a,b) if not mov[a] else (b,a)
            s1 = np.all(np.abs(d) <= E[st]+rb[mv])       # current: static world AABB + owner bounding radius
            s2 = np.all(np.abs(d) <= E[st]+E[mv])         # proposed: both world AABBs
        v=np.array([1,s1,s1 and s2,real]); cnt[k]+=v; tot+=v
print(env,"per state: sphere %.2f  +AABB(rbound) %.2f  +AABB(true ext) %.2f  real %.2f"%tuple(tot/N))
for k,v in sorted(cnt.items(), key=lambda kv:-kv[1][1]): print(f"  {k:14s} "+"  ".join(f"{x/N:5.2f}" for x in v))
print("--- moving-moving pairs by survival frequency")
freq=collections.Counter(); mind=collections.defaultdict(lambda:1e9)
gname=lambda g: (m.all_geom_names[int(m.geom_mjid[g])] or f"g{int(m.geom_mjid[g])}")+"@"+m.body_names[int(m.geom_body[g])]
for i in range(0,N,4):
    q=row[0].copy(); q[pi.ref_joint_pos_indexes]=qa[i]
    gp,gm=orc.fk(q); pd=orc.pair_dist(q)
    for (a,b) in pairs:
        if T[a]==0 or T[b]==0 or not (mov[a] and mov[b]): continue
        if np.linalg.norm(gp[a]-gp[b]) <= rb[a]+rb[b]:
            freq[(a,b)]+=1; mind[(a,b)]=min(mind[(a,b)], pd[idx[(a,b)]])
for (a,b),c in freq.most_common(16): print(f"  {c/(N/4):.2f}  {gname(a):40s} {gname(b):40s} min dist seen {mind[(a,b)]:.4f}")
