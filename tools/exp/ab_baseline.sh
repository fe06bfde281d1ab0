#!/bin/bash
# Does the oracle pool that runs before the GPU part slow the timed steps down?  Default command vs allocator set-up
# first vs a pause after the pool, alternating on one box.
show() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$1', d['ms_per_step'], 'host_exposed', d['pipeline_roofline']['host_exposed_ms_per_step'], 'zx', d['kernels']['zxpass']['ms_per_step'])
"; }
for rep in 1 2; do
python bench.py --no-sub-records 2>/dev/null | show "default"
MMX_BENCH_HEAP_EARLY=1 python bench.py --no-sub-records 2>/dev/null | show "heap early"
MMX_BENCH_HEAP_EARLY=1 MMX_BENCH_SETTLE_S=3 python bench.py --no-sub-records 2>/dev/null | show "heap early + settle 3 s"
done
