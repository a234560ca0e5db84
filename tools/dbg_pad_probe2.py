"""Developer probe: bench.py with a different hidden-width padding multiple (TunableOp tunes the new shapes in warm-up)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mult = int(sys.argv[1])
sys.argv = ["bench.py", "--no-cpu-baseline", "--warmup", "8"] + sys.argv[2:]
import bench
from torecsys_amd import layers
layers.PAD_MULTIPLE = mult
bench.main()
