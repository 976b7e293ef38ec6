"""Optional per-kernel-class HIP-event timing (used by bench.py for the live `roofline` figures).

All kernels are launched on torch's current stream, so torch.cuda.Event pairs recorded around a
launch measure that kernel on the stream it runs on.  Disabled (zero overhead) unless a `Recorder`
is active."""
from collections import defaultdict

import torch

_active = None
shape_log = None     # a list while tools/gemm_by_shape.py records one forward: every GEMM launch's [(M, N, K, epilogue), ...] in launch order


def log_gemm(dtype: str, problems):
    """One GEMM launch (one or two problems) for the per-dispatch table of tools/gemm_by_shape.py; a no-op unless that tool is recording."""
    if shape_log is not None:
        shape_log.append({"dtype": dtype, "problems": [list(map(int, p)) for p in problems if p[0] > 0]})
GEMM_KERNEL_NAMES = {0: "fdm::gemm_kernel", 1: "fdm::gemm_w4_kernel", 2: "fdm::gemm_p4_kernel", 3: "fdm::gemv_bf16_kernel",
                     4: "fdm::gemm_kernel<.., 2> (half tiles)", 5: "fdm::gemm_w4_kernel<.., 7> (224-row tiles)"}   # fdm_gemm_last_kernel()


class Recorder:
    """every: record the spans of every `every`-th step only (begin_step).  An event pair around a launch costs the GPU a marker packet on each
    side (the next kernel cannot start under the previous one's tail): ~1.1-1.4 ms of a 99 ms FLUX step, and the host ~6 us per pair, which is 8 ms
    of a host-bound eager SDXL step.  Sampling every 5th step keeps the per-launch averages (>= 5 x 57 attention launches) and cuts that cost to a fifth."""

    def __init__(self, every: int = 1):
        self.events = defaultdict(list)   # name -> [(start, stop, work, gemm kernel id or None)]
        self.every = max(1, int(every))
        self.sampled_steps = 0

    def __enter__(self):
        global _active
        _active = self
        return self

    def begin_step(self, i: int):
        """Call at the top of timed step i: spans are recorded in steps 0, every, 2 * every, ..."""
        global _active
        on = (i % self.every) == 0
        _active = self if on else None
        self.sampled_steps += 1 if on else 0

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _, _ in evs]
            out[name] = {"launches": len(evs), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms), "work": sum(e[2] for e in evs)}
            by_kernel = {}
            for t, e in zip(ms, evs):
                if e[3] is not None:
                    k = by_kernel.setdefault(GEMM_KERNEL_NAMES.get(e[3], str(e[3])), {"launches": 0, "total_ms": 0.0, "work": 0.0})
                    k["launches"] += 1
                    k["total_ms"] += t
                    k["work"] += e[2]
            if by_kernel:
                out[name]["by_kernel"] = by_kernel
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
            kern = None
            if self.name.startswith("gemm"):      # which tile kernel the dispatch policy picked for THIS launch (one span, several kernels)
                from fastdm_amd import _native as N
                kern = int(N.lib().fdm_gemm_last_kernel())
            _active.events[self.name].append((self.start, stop, self.work, kern))


def span(name: str, work: float = 0.0):
    """`with span('attn', flops): launch()`; `work` is the algorithmic flops (or bytes) of the launch."""
    return _Span(name, work)
