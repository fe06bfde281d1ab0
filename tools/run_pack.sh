for t in 2 1; do echo "== threads $t"
MMX_HOST_THREADS=$t timeout 600 python tools/steptrace.py --keep-heap --budget-gb 16 2>&1 | grep -E "step wall|tail after|more steps|StackPruner|final col" 
done
