import cProfile, pstats, sys, io, os
sys.argv = ["bench.py", "--steps", "20", "--warmup", "3", "--no-cpu-baseline"]
sys.path.insert(0, os.getcwd())
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
