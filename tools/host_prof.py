"""Developer probe: cProfile of bench.py's host side (which Python calls own the enqueue time).
usage: python tools/host_prof.py [bench args...]   e.g.  --force-sharded --steps 50"""
import sys, os, cProfile, pstats
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile(); pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
