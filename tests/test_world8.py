"""BASELINE config 5's layout at its full width on CPU: world_size 8 over gloo -- 512 candidates in
contiguous shards of 64 per rank (and an uneven 515), ONE equal-slot all-gather of the scores, every
rank ending up with the whole score vector in candidate order; a failing rank raises everywhere
instead of hanging; the ask/tell tuner on top of it.  The local evaluators are closed-form stand-ins:
what is under test is the sharding, the collective and the assembly (autompc_amd.tuning), exactly the
code the driver's 8-GPU run goes through with backend nccl."""
import os

import numpy as np
import pytest

from helpers import make_system
from test_sharded_eval import _free_port

WORLD = 8


def _score(c):                      # candidate-intrinsic: independent of shard and position
    return abs(c["sigma"] - 0.7) + 0.01 * c["horizon"] + 1e-3 * float(np.sum(np.log10(c["Q"])))


def _worker(rank, port, n, fail_rank, q):
    import torch.distributed as dist
    from autompc_amd.tuning import evaluate_sharded, random_candidates, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    cands = random_candidates(make_system(3, 2), n, seed=0)
    seen, stats = [], {}

    def local(shard, lo):
        seen.append((len(shard), lo))
        if rank == fail_rank:
            raise ValueError("device error on rank %d" % rank)
        assert shard[0] is cands[lo]
        return np.array([_score(c) for c in shard])
    try:
        scores = evaluate_sharded(local, cands, stats=stats)
        q.put((rank, "ok", scores, seen, stats.get("ranks_in_gather"), shard_bounds(n, rank, WORLD)))
    except Exception as e:          # noqa: BLE001
        q.put((rank, type(e).__name__ + ": " + str(e), None, seen, None, None))
    dist.destroy_process_group()


def _spawn(target, args_of_rank):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=args_of_rank(r) + (q,)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return {g[0]: g[1:] for g in got}


@pytest.mark.parametrize("n", [512, 515])
def test_eight_ranks_gather_every_score_in_candidate_order(n):
    from autompc_amd.tuning import random_candidates
    port = _free_port()
    got = _spawn(_worker, lambda r: (r, port, n, -1))
    ref = np.array([_score(c) for c in random_candidates(make_system(3, 2), n, seed=0)])
    sizes = []
    for r in range(WORLD):
        status, scores, seen, ranks, (lo, hi) = got[r]
        assert status == "ok" and ranks == WORLD
        np.testing.assert_array_equal(scores, ref)                 # every rank: all scores, in order
        assert seen == [(hi - lo, lo)]                              # one contiguous shard, its global offset
        sizes.append(hi - lo)
    assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    if n == 512:
        assert sizes == [64] * 8                                    # BASELINE config 5: 64 per GPU


def test_one_failing_rank_of_eight_raises_everywhere():
    port = _free_port()
    got = _spawn(_worker, lambda r: (r, port, 512, 5))
    assert got[5][0] == "ValueError: device error on rank 5"
    for r in range(WORLD):
        if r != 5:
            assert got[r][0].startswith("RuntimeError") and "rank(s) [5]" in got[r][0]


class _Formula:
    def __init__(self):
        self.calls = []

    def evaluate(self, candidates, seed=0, index_offset=0, return_trajectories=False):
        self.calls.append((len(candidates), index_offset))
        return np.array([_score(c) for c in candidates])


def _tuner_worker(rank, port, q):
    import torch.distributed as dist
    from autompc_amd.tuning import BatchPipelineTuner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    ev = _Formula()
    tuner = BatchPipelineTuner(make_system(3, 2), ev, batch_size=512, balance=False)
    best, res = tuner.run(600, np.random.default_rng(4), seed=3)
    # the default: shards balanced by work -- the same costs, every candidate evaluated exactly once
    ev2 = _Formula()
    best2, res2 = BatchPipelineTuner(make_system(3, 2), ev2, batch_size=512).run(600, np.random.default_rng(4), seed=3)
    np.testing.assert_array_equal(np.asarray(res2.costs), np.asarray(res.costs))
    q.put((rank, np.asarray(res.costs), ev.calls, res.cfgs.index(best)))
    dist.destroy_process_group()


def test_batch_tuner_over_eight_ranks():
    """600 proposals in batches of 512: batch 1 = 64 per rank, batch 2 = 88 = 11 per rank; every rank
    holds the same result."""
    port = _free_port()
    got = _spawn(_tuner_worker, lambda r: (r, port))
    costs0, _, best0 = got[0]
    for r in range(WORLD):
        costs, calls, best = got[r]
        np.testing.assert_array_equal(costs, costs0)
        assert best == best0 == int(np.argmin(costs0))
        assert calls == [(64, 64 * r), (11, 512 + 11 * r)]
    assert len(costs0) == 600


def _balanced_worker(rank, port, q):
    import torch.distributed as dist
    from autompc_amd.tuning import candidate_work, evaluate_sharded, random_candidates
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    # a batch that arrives SORTED by work (a horizon sweep, an optimiser's ranked proposals)
    cands = sorted(random_candidates(make_system(3, 2), 512, seed=0), key=lambda c: c["num_path"] * c["horizon"])
    seen, stats = [], {}

    def local(shard, ids):
        seen.append(np.asarray(ids).copy())
        assert all(shard[k] is cands[int(i)] for k, i in enumerate(np.atleast_1d(ids)))
        return np.array([_score(c) for c in shard])
    scores = evaluate_sharded(local, cands, stats=stats, weights="auto")
    work = float(sum(candidate_work(cands[int(i)]) for i in seen[0]))
    q.put((rank, scores, seen[0], work, stats["heaviest_over_mean"]))
    dist.destroy_process_group()


def test_balanced_shards_of_a_sorted_batch():
    """evaluate_sharded(weights="auto") on a work-sorted batch of 512 over 8 ranks: every candidate is
    evaluated exactly once, scores come back in candidate order, and the heaviest rank carries at most
    1.15 x the mean work (contiguous shards of the same batch: the last rank carries ~2.5 x)."""
    from autompc_amd.tuning import balanced_shards, candidate_work, random_candidates, shard_bounds
    port = _free_port()
    got = _spawn(_balanced_worker, lambda r: (r, port))
    cands = sorted(random_candidates(make_system(3, 2), 512, seed=0), key=lambda c: c["num_path"] * c["horizon"])
    ref = np.array([_score(c) for c in cands])
    all_ids = np.concatenate([np.atleast_1d(got[r][1]) for r in range(WORLD)])
    assert sorted(all_ids.tolist()) == list(range(512))
    works = np.array([got[r][2] for r in range(WORLD)])
    for r in range(WORLD):
        np.testing.assert_array_equal(got[r][0], ref)
        assert abs(got[r][3] - works.max() / works.mean()) < 1e-12
    assert works.max() <= 1.15 * works.mean()
    w = np.array([candidate_work(c) for c in cands])
    contiguous = np.array([w[slice(*shard_bounds(512, r, WORLD))].sum() for r in range(WORLD)])
    assert contiguous.max() > 1.5 * contiguous.mean()              # what balancing is for
    # the assignment is a pure function of the weights (every rank computes the same one)
    for r, ix in enumerate(balanced_shards(w, WORLD)):
        np.testing.assert_array_equal(ix, np.atleast_1d(got[r][1]))
