#!/usr/bin/env python
"""cProfile of the reference-faithful training step (tools/bench_trainstep.py) -- where the host time goes."""
import cProfile, io, os, pstats, runpy, sys
sys.argv = ["bench_trainstep.py", "12", "12"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_trainstep.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(30)
    print(s.getvalue()[:5500])
