// mopa_envdyn.hip -- second translation unit of libmopa_hip.so: K4 batched env.step (mopa_env.inc), K6 servo dynamics
// (mopa_dyn.inc), K7 contacts + constraint solver (mopa_contact.inc), rollout bookkeeping (mopa_rollstep.inc).  C ABI in include/mopa_hip.h "mopa_env_*".
// Split from mopa_hip.hip so that the dynamics kernels rebuild without the validity / planner kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mopa_hip.h"
#include "mopa_device.hpp"
#include "mopa_host.hpp"

using namespace mopa;

#include "mopa_env.inc"
#include "mopa_dyn.inc"
#include "mopa_contact.inc"
#include "mopa_rollstep.inc"
