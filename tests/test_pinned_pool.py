"""Accounting of the page-locked result-array pool (papteam_opticalflow_amd/pinned_pool.py) with a stub allocator: no GPU,
no HIP.  What the bindings rely on: the budget counts idle AND live blocks, idle blocks are evicted least recently used
first, blocks are shared per size class, and nothing calls the allocator after drain()."""
import numpy as np

from papteam_opticalflow_amd.pinned_pool import PinnedPool, size_class

MB = 1 << 20


class Stub:
    def __init__(self):
        self.next, self.live, self.freed, self.calls = 0x1000, {}, [], 0

    def alloc(self, n):
        self.calls += 1
        self.next += 0x10000000
        self.live[self.next] = n
        return self.next

    def free(self, a):
        self.freed.append(a)
        del self.live[a]  # KeyError = double free

    @property
    def pinned(self):
        return sum(self.live.values())


def test_size_classes_are_coarse_and_cover():
    for n in list(range(1, 40 * MB, 777_777)) + [8 * 1080 * 1920, 24 * 1080 * 1920, 1 << 30]:
        c = size_class(n)
        assert c >= n and c >= MB and (c - n) <= max(MB, n // 8 + 1)
    # neighbouring frame sizes share a class: 1080 x 1920 and 1072 x 1920 float64 planes
    assert size_class(8 * 1080 * 1920) == size_class(8 * 1072 * 1920)


def test_budget_counts_idle_and_live_and_evicts_lru():
    st = Stub()
    pool = PinnedPool(st.alloc, st.free, budget_bytes=64 * MB, max_idle_per_class=4, register_atexit=False)
    a = pool.take(16 * MB)
    b = pool.take(16 * MB)
    c = pool.take(24 * MB)
    assert a and b and c and pool.live_bytes == st.pinned == 56 * MB
    assert pool.take(16 * MB) is None, "live blocks alone leave no room: the caller must fall back"
    pool.give_back(*a)
    pool.give_back(*c)
    assert pool.idle_bytes == 40 * MB and pool.pinned_bytes == st.pinned == 56 * MB
    # a new class needs room: the LEAST recently used idle block (a's) goes first, then c's if still needed
    d = pool.take(20 * MB)
    assert d is not None and st.freed == [a[0]]
    assert pool.pinned_bytes == st.pinned and pool.pinned_bytes <= 64 * MB
    e = pool.take(24 * MB)  # c's block is reused, not re-allocated
    assert e[0] == c[0] and st.calls == 4
    for blk in (b, d, e):
        pool.give_back(*blk)
    assert pool.live_bytes == 0 and pool.pinned_bytes == st.pinned <= 64 * MB


def test_many_shapes_stay_inside_the_budget():
    st = Stub()
    pool = PinnedPool(st.alloc, st.free, budget_bytes=256 * MB, register_atexit=False)
    rng = np.random.default_rng(0)
    for _ in range(400):  # a caller walking many frame sizes: three result arrays per call, dropped after the call
        h, w = int(rng.integers(200, 1100)), int(rng.integers(300, 2000))
        blks = [pool.take(8 * h * w), pool.take(8 * h * w), pool.take(24 * h * w)]
        assert pool.pinned_bytes == st.pinned <= 256 * MB
        for blk in blks:
            if blk is not None:
                pool.give_back(*blk)
        assert pool.live_bytes == 0 and pool.pinned_bytes == st.pinned <= 256 * MB
    assert st.calls < 600, "1200 requests: blocks are reused across shapes through the size classes"


def test_idle_cap_per_class_and_drain():
    st = Stub()
    pool = PinnedPool(st.alloc, st.free, budget_bytes=1 << 30, max_idle_per_class=2, register_atexit=False)
    blks = [pool.take(8 * MB) for _ in range(5)]
    for blk in blks:
        pool.give_back(*blk)
    assert pool.idle_bytes == 2 * size_class(8 * MB) == st.pinned
    held = pool.take(8 * MB)
    pool.drain()
    assert st.pinned == held[1] and pool.idle_bytes == 0
    assert pool.take(8 * MB) is None, "a drained pool hands out nothing"
    n_free = len(st.freed)
    pool.give_back(*held)  # a late destructor after drain(): must not call into the allocator
    assert len(st.freed) == n_free and pool.live_bytes == 0
    pool.reopen()
    assert pool.take(8 * MB) is not None


def test_give_back_from_inside_take_does_not_deadlock():
    """a result array collected by the cyclic GC while take() holds the pool's lock on the same thread calls
    give_back() from its __del__.  Modelled by an allocator that gives a block back while take() is running: the block is
    parked, booked by the next call, and nothing blocks."""
    from papteam_opticalflow_amd.pinned_pool import PinnedPool, size_class
    freed, pool_box = [], {}
    cls = size_class(1 << 20)

    def alloc(n):
        if pool_box.get("pending"):
            pool_box["pool"].give_back(*pool_box.pop("pending"))  # re-entrant: the lock is held by take() right now
        return 0x1000 + len(freed) * 0x100 + n % 7 + 1

    pool = PinnedPool(alloc, freed.append, budget_bytes=64 << 20, register_atexit=False)
    pool_box["pool"] = pool
    first = pool.take(1 << 20)
    assert first is not None and pool.live_bytes == cls
    pool_box["pending"] = first
    second = pool.take(2 << 20)          # alloc() runs inside the lock and gives `first` back
    assert second is not None
    third = pool.take(1 << 20)           # books the parked block first, then reuses it
    assert third is not None and third[0] == first[0]
    assert pool.live_bytes == second[1] + cls and pool.idle_bytes == 0
