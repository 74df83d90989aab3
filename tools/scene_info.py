import sys; sys.path.insert(0, ".")
import os; os.environ["MOPA_DEBUG"]="1"
from mopa_rl_amd import _lib
from mopa_rl_amd.scene import planner_inputs
for env in ("SawyerPushObstacle-v0","SawyerAssemblyObstacle-v0","PusherObstacle-v0","SawyerLiftObstacle-v0"):
    pi = planner_inputs(env)
    sc = _lib.Scene(pi.model, pi.passive_joint_idx, pi.ignored_contacts, pi.spec.contact_threshold, range_=pi.spec.range)
