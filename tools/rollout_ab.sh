#!/bin/bash
# A/B runs of the asynchronous Push rollout over RolloutConfig knobs (GPU box).  One line per knob set: agent steps/s (envs stepping per
# call) of RUNS repetitions of CALLS calls each, through tools/rollout_graphs_ab.py (MOPA_BENCH_ROLLOUT=<k=v,...> overrides the config).
#   bash tools/rollout_ab.sh [-m eager|graphs|both] [-r RUNS] [-c CALLS] "planner_streams=3,planner_workgroups=128" "planner_min_job=256" ...
# Knob sets used for the measurements quoted in DESIGN.md (rounds 2-3; each was a script of its own then):
#   streams x workgroups       planner_streams={1,2,3,4} x planner_workgroups={32,64,96,128,160,256}     (3 x 128 measured best)
#   pooling policy             planner_min_job={64,128,256,512,1024} [x planner_streams, planner_job_cap=512]
#   first-phase budget         planner_first_iters={100,150,200,300,500,600}
#   chained launches           planner_chain={0,1} x planner_streams={3,4,5}
MODE=eager; RUNS=2; CALLS=200
while getopts "m:r:c:" o; do case $o in m) MODE=$OPTARG;; r) RUNS=$OPTARG;; c) CALLS=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
cd "$(dirname "$0")/.."
for knobs in "$@"; do
  for g in ONLY_EAGER ONLY_GRAPHS; do
    if [ $MODE = eager ] && [ $g = ONLY_GRAPHS ]; then continue; fi
    if [ $MODE = graphs ] && [ $g = ONLY_EAGER ]; then continue; fi
    out=""
    for r in $(seq $RUNS); do
      v=$(env $g=1 MOPA_BENCH_ROLLOUT=$knobs python tools/rollout_graphs_ab.py 4096 $CALLS 2>&1 | grep "^graphs" | sed -e "s/.*agent_steps_per_s': \([0-9.]*\).*envs_stepping_per_call': \([0-9.]*\).*/\1 \2/" | awk '{printf "%d(%d)", $1, $2}')
      out="$out $v"
    done
    echo "$knobs [$g]: agent steps/s (envs stepping per call), $RUNS runs of $CALLS calls:$out"
  done
done
