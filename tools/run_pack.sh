ZX_CHECK_MODE=7 timeout 600 python tools/zx4_check.py 2>&1 | tail -1
ZX_CHECK_MODE=6 timeout 600 python tools/zx4_check.py 2>&1 | tail -1
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -3
