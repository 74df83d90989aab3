#!/bin/bash
# A/B helper (GPU box): bench every ab_*.so in the repo root (selected through MOPA_HIP_LIB)
for v in ab_*.so; do
  export MOPA_HIP_LIB=$PWD/$v
  python bench.py --no-cpu --steps ${STEPS:-20} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), 'M/s', round(d['roofline']['kernel_ms'],4), 'ms')"
done
