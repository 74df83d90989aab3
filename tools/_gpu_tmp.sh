timeout 2400 bash tools/rollout_ab.sh -r 4 -c 300 "planner_first_iters=150" "planner_first_iters=200" "planner_first_iters=250" "planner_first_iters=300" 2>&1 | grep -v amdgpu
