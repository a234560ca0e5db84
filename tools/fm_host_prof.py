import sys, os, cProfile, pstats
sys.argv = ["bench.py", "--no-cpu-baseline", "--model", "fm", "--steps", "50", "--warmup", "5"]
sys.path.insert(0, "/root/repo")
import bench
pr = cProfile.Profile(); pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
