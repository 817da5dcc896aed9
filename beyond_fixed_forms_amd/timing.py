"""HIP-event timers around individual kernel launches (events are recorded on torch's current stream,
which is the stream every bff kernel is launched on, see _lib._stream)."""
from __future__ import annotations

import contextlib
from collections import defaultdict

import torch


class KernelTimers:
    def __init__(self, reserve: int = 0):
        """reserve: native events created up front (hipEventCreate costs ~0.2 ms apiece the first few hundred times;
        a timed loop must not pay for it: bench.py reserves 4 per step before the clock starts)."""
        self._events = defaultdict(list)
        self._pool = []
        if reserve:
            from . import _lib
            lib = _lib.load()
            for _ in range(reserve):
                e = lib.bff_event_create()
                if not e:
                    raise RuntimeError("bff_event_create failed")
                self._pool.append(e)

    def _native_event(self):
        if self._pool:
            return self._pool.pop()
        from . import _lib
        e = _lib.load().bff_event_create()
        if not e:
            raise RuntimeError("bff_event_create failed")
        return e

    @contextlib.contextmanager
    def span(self, name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self._events[name].append((a, b))

    @contextlib.contextmanager
    def sweep_span(self, name):
        """Span of the projection sweep: the events ride on the kernel dispatch itself (bff_profile_next_sweep),
        so the elapsed time is the kernel's own duration, not launch + two event packets."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        a, b = self._native_event(), self._native_event()
        lib.bff_profile_next_sweep(a, b)
        yield
        self._events[name].append((ctypes.c_void_p(a), ctypes.c_void_p(b)))

    @contextlib.contextmanager
    def merge_span(self, name):
        """Span of the components' tile pass (bff_profile_next_merge): the kernel's own duration."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        a, b = self._native_event(), self._native_event()
        lib.bff_profile_next_merge(a, b)
        yield
        self._events[name].append((ctypes.c_void_p(a), ctypes.c_void_p(b)))

    def summary(self):
        """{name: (launches, total_ms, mean_ms)} -- call after a device synchronize."""
        import ctypes
        out = {}
        for k, evs in self._events.items():
            ms = []
            for a, b in evs:
                if isinstance(a, ctypes.c_void_p):
                    from . import _lib
                    v = ctypes.c_float(0.0)
                    if _lib.load().bff_event_elapsed_ms(a, b, ctypes.byref(v)) != 0:
                        raise RuntimeError(_lib.load().bff_last_error().decode())
                    ms.append(float(v.value))
                else:
                    ms.append(a.elapsed_time(b))
            out[k] = (len(ms), sum(ms), sum(ms) / len(ms))
        return out

    def clear(self):
        import ctypes
        for evs in self._events.values():
            for a, b in evs:
                if isinstance(a, ctypes.c_void_p):
                    from . import _lib
                    _lib.load().bff_event_destroy(a)
                    _lib.load().bff_event_destroy(b)
        self._events.clear()
        for e in self._pool:
            from . import _lib
            _lib.load().bff_event_destroy(e)
        self._pool = []


@contextlib.contextmanager
def span(timers, name):
    if timers is None:
        yield
    else:
        with timers.span(name):
            yield


@contextlib.contextmanager
def sweep_span(timers, name):
    if timers is None:
        yield
    else:
        with timers.sweep_span(name):
            yield


@contextlib.contextmanager
def merge_span(timers, name):
    if timers is None:
        yield
    else:
        with timers.merge_span(name):
            yield
