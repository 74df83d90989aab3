for mc in 8 7 8 7 8 7; do
echo "maxcon $mc: $(CT_ENVS=Lift timeout 300 python tools/ct_bench.py 4096 10 $mc 2>&1 | grep -v amdgpu | grep 'env.step' | sed -e 's/.*scale \([0-9.]*\): *\([0-9.]*\) ms.*/\1:\2ms/' | tr '\n' ' ')"
done
