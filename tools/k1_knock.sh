#!/bin/bash
# Where K1's time goes: knock-in / knock-out builds of the lane-per-state kernel, each timed on the
# bench batch (tools/scene_bench.py).  A phase that runs twice costs its own time once more; a phase left out shows what
# is left.  Run the build part in the build container, the timing part on the GPU box:
#   bash tools/k1_knock.sh build ; gpurun -- 'bash tools/k1_knock.sh time'
set -e
cd "$(dirname "$0")/.."
SRC=mopa_rl_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -Wno-unused-function"
VARIANTS="fk2:-DMOPA_V5_FK_REPS=2 cull2:-DMOPA_V5_CULL_REPS=2 nodrain:-DMOPA_V5_KO_DRAIN nopassb:-DMOPA_V5_KO_DRAIN,-DMOPA_V5_KO_PASSB"
if [ "$1" = build ]; then
    for v in $VARIANTS; do
        /opt/rocm/bin/hipcc $FLAGS $(echo ${v#*:} | tr , ' ') -c -o /tmp/mopa_knock_${v%%:*}.o $SRC/mopa_hip.hip
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $SRC/libmopa_knock_${v%%:*}.so /tmp/mopa_knock_${v%%:*}.o $SRC/mopa_envdyn.o   # (the env / dynamics TU as built by make)
    done
else
    PAT=${2:-SawyerPush}
    echo "baseline:"; python tools/scene_bench.py 2>&1 | grep -E "$PAT"
    for v in $VARIANTS; do
        echo "${v%%:*}:"; MOPA_HIP_LIB=$PWD/$SRC/libmopa_knock_${v%%:*}.so python tools/scene_bench.py 2>&1 | grep -E "$PAT"
    done
fi
