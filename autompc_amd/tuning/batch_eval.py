"""Batched, multi-GPU evaluation of tuning candidates against a surrogate model.

What it replaces.  The reference's tuner evaluates one candidate at a time:
``PipelineTuner.run -> eval_cfg(cfg)`` (autompc/tuning/pipeline_tuner.py:213-258) builds a
controller, ``reset()``s it, runs ``simulate(controller, init_obs, sim_model=surrogate,
max_steps=task.get_num_steps())`` and scores the trajectory with ``task.get_cost()(traj)``.
Here a whole batch of candidates -- each an MPPI hyper-parameter set (horizon, sigma, lmda,
num_path; mppi.py:52-63) plus QuadCost weights (quad_cost_factory.py:64-95) -- is evaluated
at once: every control step is ONE rollout launch covering all candidates' samples, the
surrogate step, the trajectory bookkeeping and the task-cost score (quadratic, threshold, box
and sums of those: ampc_mppi_closed_loop_scored) stay on the device, and only the scores come
back.

Multi-GPU.  Candidates are independent, so they are partitioned into contiguous shards, one per
rank (one process per GPU, torch.distributed).  There is no collective on the data path; the only
exchange is one all-gather of the per-candidate scores at the end (RCCL over xGMI when the
backend is "nccl"; 8 bytes per candidate, latency-bound).
"""
import numpy as np

from .. import _lib
from ..control.mppi import _quad_cost_blocks
from ..costs.terms import cost_terms


def shard_bounds(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _as_matrix(v, n):
    v = np.asarray(v, dtype=np.float64)
    return np.diag(v) if v.ndim == 1 else v.reshape(n, n)


def score_trajectories(cost, obs, ctrls):
    """``cost(traj)`` for a batch of trajectories obs [B,T+1,no], ctrls [B,T+1,nu]
    (Cost.__call__, cost.py:27-41): stage cost of every row plus the terminal cost of the last
    observation.  Quadratic costs are scored in closed form; anything else (threshold costs)
    falls back to the cost object's own per-step interface."""
    B = obs.shape[0]
    if getattr(cost, "is_quad", False):
        Q, R, F = cost.get_cost_matrices()
        d = obs - cost.get_goal()
        s = np.einsum("bti,ij,btj->b", d, Q, d) + np.einsum("bti,ij,btj->b", ctrls, R, ctrls)
        return s + np.einsum("bi,ij,bj->b", d[:, -1], F, d[:, -1])
    out = np.zeros(B)
    for b in range(B):
        for t in range(obs.shape[1]):
            out[b] += cost.eval_obs_cost(obs[b, t]) + cost.eval_ctrl_cost(ctrls[b, t])
        out[b] += cost.eval_term_obs_cost(obs[b, -1])
    return out


def _task_goal(cost, obs_dim):
    """Goal the candidates' quadratic costs are centred on: the task cost's goal (as
    QuadCostFactory takes it, quad_cost_factory.py:64-66); for a sum without a shared goal the
    first term that has one; the origin if none has."""
    if getattr(cost, "is_quad", False):
        return _quad_cost_blocks(cost)[3]
    for c in [cost] + list(getattr(cost, "costs", [])):
        try:
            g = np.asarray(c.get_goal(), dtype=np.float64)
        except Exception:
            continue
        if g.shape == (obs_dim,):
            return g.copy()
    return np.zeros(obs_dim)


class CandidateEvaluator:
    """Evaluates MPPI + QuadCost candidates for one (system, task, model, surrogate) on one GPU."""

    def __init__(self, system, task, model, surrogate=None, precision="f64", device=0,
                 tile_rows=32, horizon_cap=30):
        """tile_rows / horizon_cap fix the rollout geometry (MppiPlan.set_geometry) so that a
        candidate's score is bit-identical for any batch it is evaluated in; horizon_cap defaults
        to the top of the reference's MPPI horizon range (mppi.py:52-55)."""
        self.tile_rows, self.horizon_cap = int(tile_rows), int(horizon_cap)
        if not hasattr(model, "stage_into"):
            raise TypeError("needs a device-stageable model (autompc_amd.sysid.MLP)")
        self.system, self.task, self.model = system, task, model
        self.surrogate = surrogate if surrogate is not None else model
        self.precision, self.device = precision, device
        b = task.get_ctrl_bounds()
        self.umin, self.umax = b[:, 0].copy(), b[:, 1].copy()
        self.goal = _task_goal(task.get_cost(), system.obs_dim)

    def evaluate(self, candidates, n_steps=None, seed=0, init_obs=None, eps_all=None,
                 act_init=None, return_trajectories=False, index_offset=0, timing=None):
        """Closed-loop score of every candidate (a list of dicts with keys horizon, sigma, lmda,
        num_path, Q, R, F -- Q/R/F either diagonals or full matrices).

        All randomness of candidate i is keyed by (seed, index_offset + i): its warm start comes
        from ``default_rng([seed, index_offset + i])`` and its device noise from the Philox
        stream of that global index.  A rank that evaluates the shard ``all[lo:hi]`` with
        ``index_offset=lo`` therefore returns exactly the scores a single process returns for
        those candidates (the per-candidate ``surr_cost`` of pipeline_tuner.py:213-258).

        timing: an optional dict; on return timing["timing"] holds the average HIP-event duration of
        the rollout / update kernels over the closed loop's control steps (bench.py roofline leg)."""
        B = len(candidates)
        if B == 0:
            return (np.zeros(0), None, None) if return_trajectories else np.zeros(0)
        opened = []                       # device objects, closed on every exit path
        try:
            return self._evaluate(candidates, n_steps, seed, init_obs, eps_all, act_init,
                                  return_trajectories, int(index_offset), opened, timing)
        finally:
            for obj in reversed(opened):
                obj.close()

    def _evaluate(self, candidates, n_steps, seed, init_obs, eps_all, act_init, return_trajectories,
                  index_offset, opened, timing=None):
        nx, nu, no = self.model.state_dim, self.system.ctrl_dim, self.system.obs_dim
        B = len(candidates)
        n_steps = int(n_steps if n_steps is not None else self.task.get_num_steps())
        init_obs = self.task.get_init_obs() if init_obs is None else np.asarray(init_obs)
        h = _lib.Handle(self.device, self.precision)
        opened.append(h)
        self.model.stage_into(h)
        Q = np.stack([_as_matrix(c["Q"], no) for c in candidates])
        R = np.stack([_as_matrix(c["R"], nu) for c in candidates])
        F = np.stack([_as_matrix(c["F"], no) for c in candidates])
        h.set_quad_costs(Q, R, F, np.tile(self.goal, (B, 1)))
        h.set_ctrl_bounds(self.umin, self.umax)
        sur = None
        if self.surrogate is not self.model:
            sur = _lib.Handle(self.device, self.precision)
            opened.append(sur)
            self.surrogate.stage_into(sur)
        Hs = [int(c["horizon"]) for c in candidates]
        plan = _lib.MppiPlan(h, [int(c["num_path"]) for c in candidates], Hs,
                             [float(c["sigma"]) for c in candidates],
                             [float(c["lmda"]) for c in candidates], cost_index=np.arange(B))
        opened.append(plan)
        plan.set_geometry(self.tile_rows, self.horizon_cap)
        plan.set_noise_ids(index_offset + np.arange(B))
        if act_init is None:
            # MPPI.__init__ / reset() draw the warm start ~ N(0, sigma) (mppi.py:97-99); here from
            # a stream seeded by (seed, global candidate index)
            act_init = np.concatenate([
                np.random.default_rng([int(seed), index_offset + i]).normal(
                    scale=np.sqrt(c["sigma"]), size=Hs[i] * nu)
                for i, c in enumerate(candidates)])
        plan.upload(act_seq=act_init)
        if timing is not None:
            plan.set_timing(True)
        try:
            terms = cost_terms(self.task.get_cost(), no, nu)
        except TypeError:
            terms = None                  # a user-defined cost object: score it through its own
        if terms is not None:             # Python interface from the downloaded trajectories
            res = plan.closed_loop_scored(np.tile(init_obs, (B, 1)), n_steps, terms, seed=seed,
                                          eps_all=eps_all, surrogate=sur,
                                          return_trajectories=return_trajectories)
            scores, obs, ctrls = res if return_trajectories else (res, None, None)
        else:
            obs, ctrls = plan.closed_loop(np.tile(init_obs, (B, 1)), n_steps, seed=seed,
                                          eps_all=eps_all, surrogate=sur)
            scores = score_trajectories(self.task.get_cost(), obs[:, :, :no], ctrls)
        if timing is not None:
            timing["timing"] = plan.timing()
        return (scores, obs, ctrls) if return_trajectories else scores


def evaluate_sharded(local_eval, candidates, rank=None, world=None, device=None):
    """Score ``candidates`` with ``local_eval(sub_list, lo) -> scores`` on this rank's contiguous
    shard ``candidates[lo:hi]`` and all-gather the scores so every rank returns the full vector
    (candidate order).  ``lo`` is the global index of the shard's first candidate: an evaluator that
    keys its randomness by it (CandidateEvaluator.evaluate(..., index_offset=lo)) returns the same
    score for a candidate whatever the world size.
    Uses the default torch.distributed process group when one is initialised; with none (or world
    size 1) it is a plain local evaluation.

    A rank whose local evaluation raises still takes part in the all-gather (its slot carries a
    failure marker), so the other ranks are not left waiting inside the collective; afterwards
    every rank raises."""
    import torch
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if world > 1 else 0
    n = len(candidates)
    lo, hi = shard_bounds(n, rank, world)
    if world == 1:
        return np.asarray(local_eval(candidates[lo:hi], lo), dtype=np.float64)
    error = None
    try:
        local = np.asarray(local_eval(candidates[lo:hi], lo), dtype=np.float64)
        if local.shape != (hi - lo,):
            raise ValueError("local_eval returned %r scores for %d candidates" % (local.shape, hi - lo))
    except Exception as e:           # noqa: BLE001 -- reported after the collective
        error, local = e, np.zeros(hi - lo)
    per = (n + world - 1) // world                     # equal-sized slots for the all-gather
    dev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl"
        else torch.device("cpu"))
    slot = torch.full((per + 1,), float("nan"), dtype=torch.float64, device=dev)
    slot[:hi - lo] = torch.from_numpy(local).to(dev)
    slot[per] = 0.0 if error is None else 1.0          # failure marker of this rank
    gathered = torch.empty(world * (per + 1), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, slot)
    g = gathered.cpu().numpy().reshape(world, per + 1)
    if error is not None:
        raise error
    failed = [r for r in range(world) if g[r, per] != 0.0]
    if failed:
        raise RuntimeError("candidate evaluation failed on rank(s) %s" % failed)
    out = np.empty(n)
    for r in range(world):
        a, b = shard_bounds(n, r, world)
        out[a:b] = g[r, :b - a]
    return out


def random_candidates(system, n, seed=0):
    """Candidates drawn from the reference's config ranges: MPPI horizon 5-30, sigma 1e-4-2,
    lmda 0.1-2, num_path 100-1000 (mppi.py:52-63); QuadCost diagonal gains log-uniform in
    [1e-3, 1e4] (quad_cost_factory.py:46-58)."""
    rng = np.random.default_rng(seed)
    no, nu = system.obs_dim, system.ctrl_dim
    out = []
    for _ in range(n):
        out.append(dict(horizon=int(rng.integers(5, 31)), sigma=float(rng.uniform(1e-4, 2.0)),
                        lmda=float(rng.uniform(0.1, 2.0)), num_path=int(rng.integers(100, 1001)),
                        Q=10 ** rng.uniform(-3, 4, size=no), R=10 ** rng.uniform(-3, 4, size=nu),
                        F=10 ** rng.uniform(-3, 4, size=no)))
    return out
