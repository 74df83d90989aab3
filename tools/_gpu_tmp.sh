timeout 600 python -m pytest tests/test_gpu_dyn.py -x -q -k "refuses or noslip or chunked" 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
