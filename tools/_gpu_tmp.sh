for g in "" 1 "" 1 "" 1; do
echo "gc-control '$g': $(CT_GC=$g timeout 300 python tools/ct_bench.py 4096 10 8 2>&1 | grep -v amdgpu | grep 'env.step' | sed -e 's/Sawyer\([A-Za-z]*\)Obstacle.*scale \([0-9.]*\): *\([0-9.]*\) ms.*/\1 \2:\3/' | tr '\n' ' ')"
done
