#!/usr/bin/env python3
"""cProfile of bench.py's host side for one workload: where the Python time of a step goes.

    python tools/benchprof.py --config c2 --steps 200 --warmup 20 [--top 45]

Runs ``bench.py`` in-process (same arguments, ``--no-cpu-baseline`` added) under cProfile and prints the functions by
own time and by cumulative time.  The profiler inflates Python-heavy code by 1.3-2 x: read the distribution, not the
total."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

top = 45
argv = sys.argv[1:]
if "--top" in argv:
    i = argv.index("--top")
    top = int(argv[i + 1])
    del argv[i:i + 2]
sys.argv = ["bench.py"] + argv + ["--no-cpu-baseline"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).strip_dirs().sort_stats(key).print_stats(top)
    print(buf.getvalue()[:12000], file=sys.stderr)
