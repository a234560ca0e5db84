"""Developer probe: does torch.cuda.synchronize() wait for queued work?  (wall clock vs event span of a 40 ms queue)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
def run(tag, pre=None, with_events=False):
    for _ in range(3): a @ b
    torch.cuda.synchronize()
    if pre: pre()
    e0 = e1 = None
    if with_events:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    t = time.perf_counter()
    for _ in range(50): a @ b
    if with_events: e1.record()
    enq = time.perf_counter() - t
    torch.cuda.synchronize()
    wall = time.perf_counter() - t
    # ground truth: a second, event-based measurement of the same queue length
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(50): a @ b
    s1.record(); s1.synchronize()
    print("%-34s enqueue %.2f ms  wall-after-synchronize %.2f ms  (event span of the same work %.2f ms)%s" %
          (tag, enq * 1e3, wall * 1e3, s0.elapsed_time(s1), "" if not with_events else "  in-region events %.2f ms" % e0.elapsed_time(e1)))
run("plain")
run("plain, events in region", with_events=True)
from torecsys_amd import _abi, functional as F_
x = torch.randn(4096, 8, 64, device=dev)
def trs():
    fm = torch.empty(4096, 64, device=dev); s = torch.empty(4096, 64, device=dev)
    _abi.call("trs_fm_fwd", _abi.ptr(x), 4096, 8, 64, 0, _abi.ptr(fm), _abi.ptr(s), _abi.stream_ptr())
run("after a libtrs launch", pre=trs)
side = torch.cuda.Stream()
def games():
    main = torch.cuda.current_stream(0)
    side.wait_stream(main); torch.cuda.set_stream(side); trs(); ev = torch.cuda.Event(); ev.record(side); torch.cuda.set_stream(main); main.wait_event(ev)
run("after set_stream side/main", pre=games)
import gc
gc.disable(); run("gc disabled"); gc.enable()
