"""Developer probe: who calls torch.cuda.is_available() / device_count() inside a training step?"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
stacks = collections.Counter()
orig = torch._C._cuda_getDeviceCount
def wrapped():
    st = traceback.extract_stack(limit=7)[:-1]
    stacks[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}:{f.name}" for f in reversed(st))] += 1
    return orig()
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"]
torch._C._cuda_getDeviceCount = wrapped
bench.main()
for k, v in stacks.most_common(8):
    print(v, k)
