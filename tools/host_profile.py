"""cProfile of the engine's host side over a few bench steps (GPU box): where the Python time of a step goes."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "32", "--warmup", "8"]
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
