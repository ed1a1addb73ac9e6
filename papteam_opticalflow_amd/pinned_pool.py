"""Accounting for the page-locked RESULT arrays of the Python bindings (include/papof.h: papof_host_alloc).

The reference's `pyflow.pyx` allocates vx, vy, warpI2 with np.zeros for every call (Code/Serial/pyflow.pyx:44-52); the
bindings here hand out recycled page-locked blocks instead (direct DMA, no first-touch faults).  Page-locked memory is
memory the OS cannot page, so the pool is BOUNDED and its bound counts everything it has pinned -- the blocks callers
hold AND the idle ones waiting for reuse:

  * blocks are kept per SIZE CLASS (eight classes per power of two, <= 12.5 % slack), not per exact byte count, so a
    caller that walks many frame sizes reuses blocks instead of pinning a new set per shape;
  * `take()` evicts idle blocks, least recently used first, until live + idle + the new block fit the budget; when the
    live blocks alone leave no room it returns None and the binding falls back to plain np.zeros;
  * `give_back()` keeps at most `max_idle_per_class` idle blocks of a class and evicts (LRU) beyond the budget;
  * `drain()` frees every idle block; it is registered with `atexit`, and once it has run a returning block is simply
    dropped (at interpreter shutdown the HIP runtime may already be gone: never call into it from a late destructor).

Pure Python with the allocator passed in, so that the accounting is unit-tested on a CPU-only box with a stub allocator
(tests/test_pinned_pool.py).  Used by capi.py (ctypes) and by the Cython drop-in (dropin/pyflow.pyx loads this file by
path: the drop-in directory alone is on the caller's sys.path).
"""
import atexit
import os
import threading


def size_class(nbytes):
    """Smallest class size >= nbytes: multiples of 2^(k-3) inside [2^k, 2^(k+1)), at least 1 MiB granules."""
    nbytes = int(nbytes)
    if nbytes <= (1 << 20):
        return 1 << 20
    k = nbytes.bit_length() - 1            # 2^k <= nbytes
    step = max(1 << 20, 1 << (k - 3))
    return ((nbytes + step - 1) // step) * step


def default_budget():
    try:
        return int(float(os.environ.get("PAPOF_PINNED_BUDGET_MB", "1024")) * (1 << 20))
    except ValueError:
        return 1 << 30


class PinnedPool(object):
    def __init__(self, alloc, free, budget_bytes=None, max_idle_per_class=4, register_atexit=True):
        """alloc(nbytes) -> address (int) or 0 / None on failure; free(address)."""
        self._alloc, self._free = alloc, free
        self.budget = default_budget() if budget_bytes is None else int(budget_bytes)
        self.max_idle = int(max_idle_per_class)
        self.live_bytes = 0          # handed out, not yet given back
        self.idle_bytes = 0          # pinned, waiting for reuse
        self._idle = {}              # class size -> [(tick, address), ...] oldest first
        self._tick = 0
        self._closed = False
        # give_back() runs from a result array's __del__, which the cyclic GC may call while take() holds the lock ON THIS VERY
        # THREAD (take() allocates Python objects): it never blocks -- a block it cannot book at once is parked in _deferred
        # (list.append is atomic) and booked by whoever holds the lock next
        self._lock = threading.Lock()
        self._deferred = []
        if register_atexit:
            atexit.register(self.drain)

    # ---- accounting -------------------------------------------------------------------------------------
    @property
    def pinned_bytes(self):
        return self.live_bytes + self.idle_bytes

    def _evict_lru(self, need_room):
        """free idle blocks, least recently used first, until `need_room` more bytes fit the budget (or none are left)"""
        while self.pinned_bytes + need_room > self.budget and self.idle_bytes > 0:
            cls = min((c for c, lst in self._idle.items() if lst), key=lambda c: self._idle[c][0][0])
            _, addr = self._idle[cls].pop(0)
            self.idle_bytes -= cls
            self._free(addr)

    def take(self, nbytes):
        """(address, class_bytes) of a pinned block of at least nbytes, or None (budget exhausted / allocator failed /
        pool drained): the caller then uses ordinary memory."""
        cls = size_class(nbytes)
        with self._lock:
            self._book_deferred()
            if self._closed or cls > self.budget:
                return None
            lst = self._idle.get(cls)
            if lst:
                _, addr = lst.pop()  # most recently used block of the class: warmest
                self.idle_bytes -= cls
                self.live_bytes += cls
                return addr, cls
            self._evict_lru(cls)
            if self.pinned_bytes + cls > self.budget:
                return None
            addr = self._alloc(cls)
            if not addr:
                return None
            self.live_bytes += cls
            return addr, cls

    def _book(self, addr, cls):
        """(lock held) a block comes back: keep it idle, or free it"""
        self.live_bytes -= cls
        if self._closed:
            return  # after drain(): the runtime may be gone, the process is ending -- drop, do not call into it
        lst = self._idle.setdefault(cls, [])
        if len(lst) >= self.max_idle:
            self._free(addr)
            return
        self._tick += 1
        lst.append((self._tick, addr))
        self.idle_bytes += cls
        self._evict_lru(0)

    def _book_deferred(self):
        while self._deferred:
            addr, cls = self._deferred.pop()
            self._book(addr, cls)

    def give_back(self, addr, cls):
        if not self._lock.acquire(blocking=False):
            self._deferred.append((addr, cls))
            return
        try:
            self._book(addr, cls)
            self._book_deferred()
        finally:
            self._lock.release()

    def drain(self):
        """free every idle block and stop pooling (atexit; also callable by a host that wants the memory back)"""
        with self._lock:
            self._book_deferred()
            if self._closed:
                return
            self._closed = True
            for cls, lst in self._idle.items():
                for _, addr in lst:
                    try:
                        self._free(addr)
                    except Exception:  # noqa: BLE001 -- shutting down
                        pass
            self._idle.clear()
            self.idle_bytes = 0

    def reopen(self):
        """pool again after a drain() (tests, long-lived hosts)"""
        with self._lock:
            self._closed = False
