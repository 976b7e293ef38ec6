"""Optional per-kernel-class HIP-event timing (used by bench.py for the live `roofline` figures).

All kernels are launched on torch's current stream, so torch.cuda.Event pairs recorded around a
launch measure that kernel on the stream it runs on.  Disabled (zero overhead) unless a `Recorder`
is active."""
from collections import defaultdict

import torch

_active = None


class Recorder:
    def __init__(self):
        self.events = defaultdict(list)   # name -> [(start, stop, work)]

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            out[name] = {"launches": len(evs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms), "work": sum(w for _, _, w in evs)}
        return out


class _Span:
    __slots__ = ("name", "work", "start")

    def __init__(self, name, work):
        self.name, self.work = name, work

    def __enter__(self):
        if _active is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if _active is not None:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record()
            _active.events[self.name].append((self.start, stop, self.work))


def span(name: str, work: float = 0.0):
    """`with span('attn', flops): launch()`; `work` is the algorithmic flops (or bytes) of the launch."""
    return _Span(name, work)
