MOPA_CT_PAIR_SORT=1 timeout 900 python -m pytest tests/test_gpu_dyn.py -x -q -k "joint_limit_rows" 2>&1 | grep -E "^E|assert|Error" | head -12
