"""cProfile of tools/bench_nobrs.py (host side of the NoBRS click loop)."""
import cProfile, pstats, sys, io, os, runpy
sys.argv = ["bench_nobrs.py", "20"]
sys.path.insert(0, os.getcwd())
pr = cProfile.Profile()
pr.enable()
runpy.run_path("tools/bench_nobrs.py", run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
