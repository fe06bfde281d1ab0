#!/bin/bash
# A/B of two builds of libmmx_hip.so on one box: bench.py lines (c3 + appended c2 / c5) and the c3 step trace, alternating.
# The other build goes to magellanmapper_amd/libmmx_old.so first (git stash; make -C magellanmapper_amd/csrc; cp ../libmmx_hip.so
# ../libmmx_old.so; git stash pop; make again) and is selected through MMX_LIB_PATH.
OLD=$PWD/magellanmapper_amd/libmmx_old.so
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export MMX_LIB_PATH=$OLD; else unset MMX_LIB_PATH; fi
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        also = [(k, d[k].get('ms_per_step')) for k in d if isinstance(d[k], dict) and 'ms_per_step' in d[k] and k != 'roofline']
        print('$lib', d['ms_per_step'], d['table_sha1'][:8], 'host_exposed', d['pipeline_roofline']['host_exposed_ms_per_step'], also)
"
    python tools/steptrace.py --config c3 2>&1 | tail -2 | sed "s/^/$lib /"
  done
done
