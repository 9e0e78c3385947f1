"""CPU: the bounded plan caches (usdm_amd/plancache.py) — LRU order / eviction, length buckets, and the replayable arena that
lets exact-length plans of one bucket share a workspace (re-zeroed when another plan takes it over)."""
import torch

from usdm_amd.plancache import LRU, Arena, bucket


def test_bucket():
    assert [bucket(n, 32) for n in (1, 32, 33, 1118)] == [32, 32, 64, 1120]


def test_lru_evicts_least_recently_used():
    c = LRU(3)
    built = []
    for k in "abc":
        c.get_or_build(k, lambda k=k: built.append(k) or k.upper())
    assert c.get("a") == "A"                 # touch a: b is now the oldest
    c.get_or_build("d", lambda: "D")
    assert "b" not in c and list(c) == ["c", "a", "d"] and len(c) == 3 and c.evictions == 1
    assert c.get_or_build("a", lambda: 1 / 0) == "A"     # cached: builder not called
    assert built == ["a", "b", "c"]


def test_arena_replays_the_same_buffers_and_rezeroes_on_owner_switch():
    a = Arena("cpu")
    a.begin()
    big = [a.zeros(8, 4), a.zeros(10, dtype=torch.int64), a.zeros(3, 3, 3)]          # reservation at the bucket's capacity
    n0 = a.nbytes()
    a.begin()
    p1 = [a.zeros(6, 4), a.zeros(7, dtype=torch.int64), a.zeros(2, 3, 3)]
    a.begin()
    p2 = [a.zeros(8, 4), a.zeros(2, dtype=torch.int64), a.zeros(1, 3, 3)]
    assert a.nbytes() == n0                                                           # nothing new was allocated
    assert all(x.data_ptr() == y.data_ptr() == z.data_ptr() for x, y, z in zip(big, p1, p2))
    assert p1[0].shape == (6, 4) and p2[2].shape == (1, 3, 3)
    a.take("p1")
    p1[0].fill_(7.0)
    a.take("p1")
    assert float(p1[0].sum()) == 7.0 * 24                                             # same owner: contents kept
    a.take("p2")
    assert float(p2[0].sum()) == 0.0                                                  # other owner: back to all-zero
    # a request larger than the reservation gets its own buffer instead of overrunning
    a.begin()
    q = a.zeros(100, 4)
    assert q.numel() == 400 and a.nbytes() > n0


def test_arena_keeps_replaced_buffers_alive_and_zeroes_them_too():
    """plans record raw pointers, so a buffer that a later, larger (or other-dtype) build replaces must stay alive -
    and be restored to all-zero on an owner switch - for as long as the arena lives."""
    a = Arena("cpu")
    a.begin()
    small = a.zeros(4, 4)
    ptr = small.data_ptr()
    a.begin()
    big = a.zeros(64, 4)                       # does not fit the 16-element reservation -> replaces it
    assert big.data_ptr() != ptr and len(a.retired) == 1 and a.retired[0].data_ptr() == ptr
    a.begin()
    other = a.zeros(8, dtype=torch.int64)      # another dtype in the same position -> replaces again
    assert len(a.retired) == 2
    a.take("p1")
    small.fill_(3.0); big.fill_(2.0)
    a.take("p2")
    assert float(small.sum()) == 0.0 and float(big.sum()) == 0.0 and int(other.sum()) == 0
    assert a.nbytes() == (16 + 256) * 4 + 8 * 8
