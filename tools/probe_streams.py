"""Which streams overlap which: the side-stream probe of engine.Trainer._concurrent, printed per candidate.
A 0.6 ms idle kernel on `a`, a 1 us kernel on `b` behind an event recorded on `a` BEFORE the idle kernel; elapsed(e0 -> e1) < 0.3 ms = concurrent."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from counting_detr_amd import _ffi


def elapsed(a, b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        e0.record(a)
        _ffi.check(_ffi.lib().cdetr_delay(600, _ffi.stream_ptr()), "cdetr_delay")
    b.wait_event(e0)
    with torch.cuda.stream(b):
        _ffi.check(_ffi.lib().cdetr_delay(1, _ffi.stream_ptr()), "cdetr_delay")
        e1.record(b)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


if __name__ == "__main__":
    torch.zeros(1, device="cuda")
    cands = [torch.cuda.Stream() for _ in range(10)]
    main = torch.cuda.current_stream()
    print("main = the default stream:", " ".join("%.3f" % elapsed(main, c) for c in cands))
    print("again                    :", " ".join("%.3f" % elapsed(main, c) for c in cands))
    m2 = cands[0]
    print("main = candidate 0       :", " ".join("%.3f" % elapsed(m2, c) for c in cands[1:]))
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(3):
        (x @ x).sum().item()
    print("after some torch work    :", " ".join("%.3f" % elapsed(main, c) for c in cands))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(main)
    with torch.cuda.graph(g, stream=s):
        y = x * 2
    g.replay()
    torch.cuda.synchronize()
    print("after a graph capture    :", " ".join("%.3f" % elapsed(main, c) for c in cands))
    c2 = [torch.cuda.Stream() for _ in range(8)]
    print("fresh candidates         :", " ".join("%.3f" % elapsed(main, c) for c in c2))
