"""Host placement of the ranks of a multi-GPU run (autompc_amd/tuning/hostpin.py): pure host logic."""
import os

import pytest

from autompc_amd.tuning import hostpin


def test_cpulist_parsing():
    assert hostpin._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert hostpin._parse_cpulist("") == []


def test_eight_ranks_on_two_numa_nodes_share_their_own_node():
    """GPUs 0-3 on node 0 (CPUs 0-63), GPUs 4-7 on node 1 (CPUs 64-127): every rank gets 16 CPUs of ITS node,
    disjoint from every other rank's."""
    allowed = list(range(128))
    nodes = {r: (r // 4, list(range(64 * (r // 4), 64 * (r // 4) + 64))) for r in range(8)}
    got = [hostpin.rank_cpus(r, 8, allowed, nodes) for r in range(8)]
    assert all(len(g) == 16 for g in got)
    assert sorted(c for g in got for c in g) == allowed
    for r, g in enumerate(got):
        assert all(c // 64 == r // 4 for c in g)


def test_without_numa_information_the_allowed_cpus_are_divided_evenly():
    got = [hostpin.rank_cpus(r, 3, list(range(10))) for r in range(3)]
    assert got == [[0, 1, 2], [3, 4, 5], [6, 7, 8, 9]]
    # more ranks than CPUs: everybody still gets one
    assert all(len(hostpin.rank_cpus(r, 8, [0, 1])) >= 1 for r in range(8))
    # a node whose CPUs this process may not use at all falls back to the allowed set
    assert hostpin.rank_cpus(0, 1, [4, 5], {0: (1, [64, 65])}) == [4, 5]


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="needs sched_setaffinity")
def test_pin_rank_applies_and_can_be_disabled(monkeypatch):
    before = os.sched_getaffinity(0)
    try:
        monkeypatch.setenv("AMPC_PIN", "0")
        assert hostpin.pin_rank(0, 2) is None and os.sched_getaffinity(0) == before
        monkeypatch.setenv("AMPC_PIN", "1")
        rec = hostpin.pin_rank(1, 2, set_thread_caps=False)
        if len(before) >= 2:
            assert rec is not None and set(rec["cpus"]) < before and os.sched_getaffinity(0) == set(rec["cpus"])
            assert hostpin.thread_cap(10 ** 6) == len(rec["cpus"])
    finally:
        os.sched_setaffinity(0, before)
