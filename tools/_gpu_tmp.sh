timeout 1800 python -m pytest tests/test_gpu_dyn.py -x -q 2>&1 | tail -3
for srt in 1 0; do
echo "sort $srt: $(MOPA_CT_PAIR_SORT=$srt timeout 300 python tools/ct_bench.py 4096 10 8 2>&1 | grep -v amdgpu | grep 'env.step' | sed -e 's/Sawyer\([A-Za-z]*\)Obstacle.*scale \([0-9.]*\): *\([0-9.]*\) ms.*/\1 \2:\3/' | tr '\n' ' ')"
done
