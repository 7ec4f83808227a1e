import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-ahds", "--profile-iters", "1"]
sys.path.insert(0, ROOT)
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
