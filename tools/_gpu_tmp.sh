timeout 2400 bash tools/profile.sh r04 > gpurun_out/prof_r04.log 2>&1
tail -25 gpurun_out/prof_r04.log
ls gpurun_out/prof_r04 | head -40
