"""HIP-event timers around individual kernel launches (events are recorded on torch's current stream,
which is the stream every bff kernel is launched on, see _lib._stream)."""
from __future__ import annotations

import contextlib
from collections import defaultdict

import torch


class KernelTimers:
    def __init__(self):
        self._events = defaultdict(list)

    @contextlib.contextmanager
    def span(self, name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self._events[name].append((a, b))

    def summary(self):
        """{name: (launches, total_ms, mean_ms)} -- call after a device synchronize."""
        out = {}
        for k, evs in self._events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[k] = (len(ms), sum(ms), sum(ms) / len(ms))
        return out

    def clear(self):
        self._events.clear()


@contextlib.contextmanager
def span(timers, name):
    if timers is None:
        yield
    else:
        with timers.span(name):
            yield
