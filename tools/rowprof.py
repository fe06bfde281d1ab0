#!/usr/bin/env python3
"""cProfile of the host side of tools/rowbench.py's step (same arguments)."""
import cProfile, pstats, sys, os, runpy, io
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ns = {}
sys.argv = ["rowbench.py"] + sys.argv[1:] + ["--steps", "1"]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rowbench.py")).read()
head, tail = src.split("one_step()\nnat.timing_enable(True)")
g = {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "rowbench.py")}
exec(compile(head, "rowbench_head", "exec"), g)
g["one_step"](); g["one_step"]()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    g["one_step"]()
g["torch"].cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
