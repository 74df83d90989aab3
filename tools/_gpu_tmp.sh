timeout 600 python tools/ct_tail.py lift 10 11 2>&1 | grep -v amdgpu | cut -c1-260
