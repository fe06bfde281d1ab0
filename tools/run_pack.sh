timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/steptrace.py --keep-heap --budget-gb 16 2>&1 | grep -E "step wall|tail after|more steps" 
