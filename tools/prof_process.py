"""cProfile of the red_buoy body loop of tools/exp_process.py (posts off): where the host time of a process() call goes."""
import cProfile, pstats, io, os, sys, runpy
sys.argv = ["exp_process.py", "150"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exp_process.py"), run_name="__main__")
finally:
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
