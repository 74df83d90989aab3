timeout 1500 bash tools/dyn_prof.sh r04_k7n contacts > gpurun_out/prof_r04_k7n.log 2>&1
tail -5 gpurun_out/prof_r04_k7n.log
ls gpurun_out/prof_r04_k7n
