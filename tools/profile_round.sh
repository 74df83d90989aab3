#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/rNN/ is built from, one call.
#   bash tools/profile_round.sh r05
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/round_$TAG
mkdir -p $O
bash tools/profile.sh $TAG > $O/profile.log 2>&1
# the validity kernel on the other two Sawyer scenes (BASELINE configs 4 / 5): trace + PMC passes per scene
MOPA_BENCH_ENV=SawyerLiftObstacle-v0 bash tools/profile.sh ${TAG}_lift k1only > $O/profile_lift.log 2>&1
MOPA_BENCH_ENV=SawyerAssemblyObstacle-v0 bash tools/profile.sh ${TAG}_assembly k1only > $O/profile_assembly.log 2>&1
make -s -C mopa_rl_amd/csrc libmopa_hip_stats.so > /dev/null 2>&1     # (present already when built in the container: no-op)
bash tools/profile_k3.sh $TAG > $O/profile_k3.log 2>&1
bash tools/dyn_prof.sh ${TAG}_k7 contacts > $O/dyn_prof_k7.log 2>&1
bash tools/dyn_prof.sh ${TAG}_k6 push > $O/dyn_prof_k6.log 2>&1
python tools/parity_sweep.py 2>&1 | grep -v amdgpu.ids > $O/parity_sweep.txt
python tools/plan_parity_sweep.py 2>&1 | grep -v amdgpu.ids > $O/plan_parity_sweep.txt
python tools/motion_parity_sweep.py 2>&1 | grep -v amdgpu.ids > $O/motion_parity_sweep.txt
python tools/ct_parity_sweep.py 1024 4 2>&1 | grep -v amdgpu.ids > $O/ct_parity_sweep.txt
python tools/count_launches.py 100 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > $O/rollout_launches_per_call.txt
python tools/ct_bench.py 4096 10 16 2>&1 | grep -v amdgpu.ids > $O/ct_bench.txt
CT_OPTS='{"cone": "pyramidal"}' python tools/ct_bench.py 4096 10 8 2>&1 | grep -v amdgpu.ids > $O/ct_bench_pyramidal.txt
python tools/dyn_lanes_ab.py 2>&1 | grep -v amdgpu.ids > $O/dyn_lanes_ab.txt
tools/ubench/bin/lone_wave > $O/lone_wave.txt 2>&1
bash tools/k7_icache.sh ${TAG}_k7ic > $O/k7_icache.txt 2>&1
python tools/k3_build_ab.py 2>&1 | grep -v amdgpu.ids > $O/k3_build_ab.txt
python tools/ct_tail.py 2>&1 | grep -v amdgpu.ids > $O/ct_tail.txt
python bench.py > $O/bench_line.json 2> $O/bench.err
cp gpurun_out/bench_full_n1.json $O/bench_line_full.json
tail -3 $O/*.txt | cut -c1-200
