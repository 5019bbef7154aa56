"""Refillable device buffers that are safe on any stream.

The layer-to-layer forms of the cost network (SCL / PSCL: include/mvsdet_hip.h) carry a zero border that is written ONCE;
the producing kernels write interior voxels only.  Such a buffer cannot come fresh from the caching allocator on every
call (it would need its border cleared every time), so the module keeps them -- and a buffer that is kept outside the
allocator is also outside the allocator's per-stream reuse tracking.  Rounds 3-4 keyed the buffers by the id of the
stream they were used on, which is correct only while stream ids are not reused and every caller stays on "its" stream.

`EventPool` replaces the key by an ordering that holds for ANY stream: every buffer carries the event recorded behind its
last use; a stream that acquires the buffer first waits for that event (a no-op on the stream that recorded it), and the
allocator is told about every foreign stream that touches the memory (`record_stream`), so dropping a buffer is safe too.
"""
from __future__ import annotations

import threading
from typing import Callable, Hashable

import torch


class _Lease:
    __slots__ = ("key", "buf", "tensors", "alloc_stream", "event")

    def __init__(self, key, buf, tensors, alloc_stream):
        self.key, self.buf, self.tensors, self.alloc_stream, self.event = key, buf, tensors, alloc_stream, None


class EventPool:
    """key -> free buffers of that key, most recently released last.  `acquire` hands one out (or makes one with `make()`
    on the current stream); `release` records an event behind everything the current stream has enqueued and returns the
    buffers to the free lists.  A buffer that is never released is simply garbage: nothing refers to it any more."""

    def __init__(self, capacity: int = 64, max_per_key: int = 4):
        self.capacity = int(capacity)
        self.max_per_key = int(max_per_key)   # free buffers of one key beyond which a busy one is waited for instead of a new one made
        self._free: dict = {}          # key -> [_Lease, ...]
        self._count = 0
        self._lock = threading.Lock()

    def __len__(self):
        return self._count

    def clear(self):
        with self._lock:
            self._free.clear()
            self._count = 0

    # pools hold device memory and events: a module that owns one pickles without it
    def __reduce__(self):
        return (EventPool, (self.capacity, self.max_per_key))

    def acquire(self, key: Hashable, make: Callable[[], object], tensors: Callable[[object], tuple], device) -> _Lease:
        cur = torch.cuda.current_stream(device)
        with self._lock:
            lst = self._free.get(key)
            lease = None
            if lst:
                # a buffer whose last use is already behind us (released on this stream, or finished): no wait at all.  One that
                # another stream is still working on is NOT taken while the key has few buffers -- two streams that run the same
                # layers side by side (the halves of CostRegNet3DGS.view_streams) would otherwise hand ONE buffer back and forth
                # and serialise on its events (measured: 8.1 instead of 7.3 ms for the network) -- a new one is made instead; only
                # a key that already owns `max_per_key` buffers waits for its oldest
                # (first a buffer this stream released itself, then one that is finished: a stream that took the OTHER stream's
                # finished buffer would leave its own to the other stream, which may find it still busy and make a third one --
                # allocations and clears of 0.4-0.75 GB inside the steady state: 14.1 instead of 11.0 ms per scene in one bench run)
                pick = next((i for i in range(len(lst) - 1, -1, -1) if lst[i].event is None or lst[i].event[0] == cur.cuda_stream), None)
                if pick is None:
                    pick = next((i for i in range(len(lst) - 1, -1, -1) if lst[i].event[1].query()), None)
                if pick is None and len(lst) >= self.max_per_key:
                    pick = 0
                if pick is not None:
                    lease = lst.pop(pick)
                    self._count -= 1
                    if not lst:
                        del self._free[key]
        if lease is None:
            buf = make()
            return _Lease(key, buf, tuple(tensors(buf)), cur.cuda_stream)
        # unconditionally -- stream ids are reused after a stream dies, so "same id" proves nothing: waiting for an event of
        # the own stream and recording the allocating stream are no-ops inside the runtime / the allocator
        if lease.event is not None:
            cur.wait_event(lease.event[1])
        for t in lease.tensors:          # the allocator must not hand the memory on while this stream still works on it
            t.record_stream(cur)
        return lease

    def release(self, leases, device) -> None:
        leases = [l for l in leases if l is not None]
        if not leases:
            return
        cur = torch.cuda.current_stream(device)
        ev = torch.cuda.Event()
        ev.record(cur)
        with self._lock:
            for lease in leases:
                lease.event = (cur.cuda_stream, ev)
                lst = self._free.pop(lease.key, [])      # re-inserted: the dict's order is the order of last release
                lst.append(lease)
                self._free[lease.key] = lst
                self._count += 1
            while self._count > self.capacity:      # varying view counts: the key released longest ago goes first
                k = next(iter(self._free))
                lst = self._free[k]
                lst.pop(0)
                self._count -= 1
                if not lst:
                    del self._free[k]
