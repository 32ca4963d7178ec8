"""Multi-rank candidate scoring on CPU: world_size 2, gloo.  The local evaluator is the oracle's
closed loop (tests may use the oracle); what is under test is the sharding, the equal-slot
all-gather and the score assembly of autompc_amd.tuning.evaluate_sharded."""
import os
import socket

import numpy as np
import pytest

from helpers import make_system
from oracle import mlp as omlp
from oracle.closed_loop import simulate
from oracle.costs import QuadCostOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle


def _candidates(n):
    rng = np.random.default_rng(5)
    return [dict(horizon=int(rng.integers(5, 9)), sigma=float(rng.uniform(0.3, 1.0)),
                 lmda=float(rng.uniform(0.3, 1.5)), num_path=int(rng.integers(20, 40)),
                 Q=rng.uniform(0.5, 2.0, size=2), R=rng.uniform(0.01, 0.1, size=1),
                 F=rng.uniform(0.5, 2.0, size=2)) for _ in range(n)]


def _oracle_eval(cands, lo=0):
    system = make_system(2, 1)
    p = omlp.random_params(2, 1, [16, 16], "tanh", seed=2)
    model = MLPOracle(system, p)
    task_cost = QuadCostOracle(np.eye(2), 0.1 * np.eye(1), np.eye(2), np.zeros(2))
    out = []
    for i, c in enumerate(cands):
        np.random.seed(int(1000 * c["sigma"]))          # candidate-intrinsic seed: shard independent
        ctl = MPPIOracle(model, QuadCostOracle(np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"]),
                                               np.zeros(2)), np.array([[-1.0, 1.0]]),
                         horizon=c["horizon"], num_path=c["num_path"], sigma=c["sigma"],
                         lmda=c["lmda"])
        obs, ctl_traj = simulate(ctl, np.array([0.3, -0.2]), model, 5)
        out.append(task_cost.traj_cost(obs, ctl_traj))
    return np.array(out)


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    from autompc_amd.tuning import evaluate_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scores = evaluate_sharded(_oracle_eval, _candidates(n))
    q.put((rank, scores))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("n", [7, 8])
def test_two_rank_gloo_matches_single_process(n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _oracle_eval(_candidates(n))
    np.testing.assert_allclose(got[0], ref, rtol=1e-12)
    np.testing.assert_allclose(got[1], ref, rtol=1e-12)


def _failing_worker(rank, world, port, q):
    import torch.distributed as dist
    from autompc_amd.tuning import evaluate_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_eval(shard, lo):
        if rank == 1:
            raise ValueError("device error on rank 1")
        return np.arange(lo, lo + len(shard), dtype=np.float64)
    try:
        evaluate_sharded(local_eval, list(range(5)))
        q.put((rank, "no error"))
    except Exception as e:           # noqa: BLE001
        q.put((rank, type(e).__name__ + ": " + str(e)))
    dist.destroy_process_group()


def test_a_failing_rank_does_not_hang_the_collective():
    """a rank that raises inside its local evaluation must still complete the
    all-gather; every rank then raises instead of blocking forever."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] == "ValueError: device error on rank 1"
    assert got[0].startswith("RuntimeError") and "rank(s) [1]" in got[0]


def test_local_eval_receives_the_shard_offset():
    from autompc_amd.tuning import evaluate_sharded, shard_bounds
    seen = []
    out = evaluate_sharded(lambda shard, lo: (seen.append((len(shard), lo)), np.zeros(len(shard)))[1],
                           list(range(9)), rank=0, world=1)
    assert seen == [(9, 0)] and out.shape == (9,)
    assert shard_bounds(9, 1, 2) == (5, 9)


def test_shard_bounds_cover_everything_once():
    from autompc_amd.tuning import shard_bounds
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_score_trajectories_matches_cost_call():
    from autompc_amd import QuadCost, ThresholdCost, Trajectory
    from autompc_amd.tuning import score_trajectories
    system = make_system(3, 2)
    rng = np.random.default_rng(0)
    obs, ctrls = rng.normal(size=(4, 6, 3)), rng.normal(size=(4, 6, 2))
    for cost in (QuadCost(system, rng.normal(size=(3, 3)), np.eye(2), rng.normal(size=(3, 3)),
                          goal=rng.normal(size=3)),
                 ThresholdCost(system, np.zeros(3), (0, 2), 0.7)):
        got = score_trajectories(cost, obs, ctrls)
        for b in range(4):
            assert abs(got[b] - cost(Trajectory(system, 6, obs[b], ctrls[b]))) < 1e-10
