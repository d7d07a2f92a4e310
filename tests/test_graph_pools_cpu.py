"""The step executor's free list of memory pools (unimm_amd/graphs.py: no graph exec is ever destroyed, the pool of a dead entry is
captured into again): which pool a new entry gets.  Pure bookkeeping -- runs without a GPU (pool handles and streams are stand-ins)."""
import itertools

import pytest
import torch


@pytest.fixture()
def pools(monkeypatch):
    from unimm_amd import graphs as G
    ids = itertools.count(1)
    monkeypatch.setattr(torch.cuda, "graph_pool_handle", lambda: ("pool", next(ids)))
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None, **k: ("stream", next(ids)))
    monkeypatch.setattr(G, "_FREE_POOLS", {})
    return G


def test_a_new_entry_prefers_the_pool_its_own_signature_left_behind(pools):
    G = pools
    dev = torch.device("cuda", 0)
    p1, s1 = G._take_pool(dev, "sigA", side=7)
    p2, s2 = G._take_pool(dev, "sigB", side=7)
    assert p1 != p2 and s1 != s2                                   # nothing free: fresh pools, fresh capture streams
    G._give_pool(0, "sigA", 7, p1, s1)
    G._give_pool(0, "sigB", 7, p2, s2)
    assert G._take_pool(dev, "sigA", side=7) == (p1, s1)           # not the most recently freed one: the same signature's
    assert G._take_pool(dev, "sigC", side=7) == (p2, s2)           # any pool of the same engine before a new one
    p3, s3 = G._take_pool(dev, "sigC", side=7)
    assert p3 not in (p1, p2) and s3 not in (s1, s2)


def test_pools_travel_with_their_capture_stream_and_prefer_the_same_engine(pools):
    G = pools
    dev = torch.device("cuda", 0)
    pa, sa = G._take_pool(dev, "sig", side=1)
    pb, sb = G._take_pool(dev, "sig", side=2)
    G._give_pool(0, "sig", 1, pa, sa)
    G._give_pool(0, "sig", 2, pb, sb)                              # freed last
    assert G._take_pool(dev, "sig", side=1) == (pa, sa)            # the engine whose image-side stream allocated into it
    assert G._take_pool(dev, "other", side=1) == (pb, sb)          # another engine's pool rather than new memory
    assert G._FREE_POOLS[0] == []
    # another device has its own list
    G._give_pool(1, "sig", 1, pa, sa)
    p, s = G._take_pool(torch.device("cuda", 0), "sig", side=1)
    assert p != pa and G._FREE_POOLS[1] == [("sig", 1, pa, sa)]
