import sys; sys.path.insert(0,'.')
from mopa_rl_amd import _lib
from mopa_rl_amd.scene import planner_inputs
for env in ["SawyerPushObstacle-v0","SawyerAssemblyObstacle-v0","PusherObstacle-v0"]:
    pi=planner_inputs(env)
    sc=_lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold)
    print(env, "v1 lds", sc.lds_bytes, "v2 lds", sc.lds_bytes + 4*(((2*7*128*8)+128*4+8+64*8+15)&~15), "pairs", sc.npair_checked)
