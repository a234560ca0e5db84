"""Developer probe: host microseconds per step by entry point / autograd Function / module (perf_counter wrappers).
usage: python tools/host_prof2.py [bench args...]"""
import sys, os, time, collections, inspect
sys.argv = ["bench.py", "--no-cpu-baseline"] + sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torecsys_amd import _abi, functional as F_, layers as L_, inputs as I_, models as M_
T = collections.defaultdict(float); Nn = collections.Counter()
def wrap(fn, key):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            T[key] += time.perf_counter() - t; Nn[key] += 1
    return w
orig_call = _abi.call
def call(name, *args):
    t = time.perf_counter()
    try:
        return orig_call(name, *args)
    finally:
        T["abi:" + name] += time.perf_counter() - t; Nn["abi:" + name] += 1
for mod in (_abi, F_, L_, I_):
    if hasattr(mod, "call"):
        mod.call = call
for mod in (F_, L_):
    for n, c in inspect.getmembers(mod, inspect.isclass):
        if issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function and c.__module__ == mod.__name__:
            c.forward = staticmethod(wrap(c.forward, f"fn:{n}.forward"))
            c.backward = staticmethod(wrap(c.backward, f"fn:{n}.backward"))
for n, c in inspect.getmembers(L_, inspect.isclass):
    if issubclass(c, torch.nn.Module) and c.__module__ == L_.__name__:
        c.forward = wrap(c.forward, f"mod:{n}")
for n, c in inspect.getmembers(I_, inspect.isclass):
    if issubclass(c, torch.nn.Module) and c.__module__ == I_.__name__:
        c.forward = wrap(c.forward, f"mod:{n}")
for n, c in inspect.getmembers(M_, inspect.isclass):
    if issubclass(c, torch.nn.Module) and c.__module__ == M_.__name__:
        c.forward = wrap(c.forward, f"mod:{n}")
_bce = wrap(torch.nn.BCEWithLogitsLoss.forward, "mod:BCEWithLogitsLoss")
_seen = [0]
def bce(self, *a, **k):
    _seen[0] += 1
    if _seen[0] == 6:          # first timed step (bench.py's default warm-up is 5 eager steps): drop the warm-up costs
        T.clear(); Nn.clear()
    return _bce(self, *a, **k)
torch.nn.BCEWithLogitsLoss.forward = bce
torch._addmm_activation = wrap(torch._addmm_activation, "aten:_addmm_activation")
F_.prefetch_row_buckets = wrap(F_.prefetch_row_buckets, "py:prefetch_row_buckets")
bench.main()
steps = Nn["mod:BCEWithLogitsLoss"]
print("timed steps:", steps)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("%-46s %8.1f us/step  (%d calls/step)" % (k, v / steps * 1e6, round(Nn[k] / steps)))
