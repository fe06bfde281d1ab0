for n in 1 2 4; do echo "== $n processes"; for r in $(seq 0 $((n-1))); do python tools/exp/gen4.py $r $n & done; wait; done
