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
from ..costs.blocks import quad_sum_block, is_quad_sum, stack_blocks
from ..costs.terms import cost_terms


def shard_bounds(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def candidate_work(c):
    """Relative device work of one candidate's episode: rollouts per control step for MPPI candidates
    (num_path x horizon), the horizon for iLQR candidates -- what evaluate_sharded balances."""
    return float(c.get("num_path", 1)) * float(c.get("horizon", 1)) if isinstance(c, dict) else 1.0


def balanced_shards(weights, world):
    """Deal items to `world` ranks so that the heaviest rank carries little more than the mean work:
    heaviest item first onto the least-loaded rank (ties: fewer items, then the lower rank).  Returns one
    ascending index array per rank; deterministic, so every rank computes the same assignment."""
    w = np.asarray(weights, dtype=np.float64)
    order = sorted(range(len(w)), key=lambda i: (-w[i], i))
    load, count, out = [0.0] * world, [0] * world, [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], count[k], k))
        out[r].append(i)
        load[r] += w[i]
        count[r] += 1
    return [np.array(sorted(ix), dtype=np.int64) for ix in out]


def model_shape_key(model):
    """What two controller models must share to run in one plan (ampc_*_plan_set_models): the class, the
    state dimension and -- MLPs -- hidden sizes and activation."""
    return (type(model).__name__, int(model.state_dim), tuple(int(v) for v in getattr(model, "hidden_sizes", ())),
            str(getattr(model, "nonlintype", "")))


def candidate_models(candidates, default):
    """(models, index): the distinct controller models the candidates carry under "model" (the evaluator's
    own for candidates without one), in order of first appearance, and each candidate's entry."""
    models, index, seen = [], [], {}
    for c in candidates:
        m = c.get("model") if isinstance(c, dict) else None
        m = default if m is None else m
        if id(m) not in seen:
            seen[id(m)] = len(models)
            models.append(m)
        index.append(seen[id(m)])
    return models, np.asarray(index, dtype=np.int32)


def model_handle(m, device, precision, opened):
    """A handle holding model `m` for a plan's model table: the model's own cached handle (staged once,
    refreshed by the model when its parameters change; the plan keeps it alive) when it lives on this
    device in this precision, else a fresh one (closed with the evaluation)."""
    if not hasattr(m, "stage_into"):
        raise TypeError("a candidate's model must be device-stageable (autompc_amd.sysid.MLP)")
    if hasattr(m, "_dev") and getattr(m, "device", None) == device and getattr(m, "precision", None) == precision:
        return m._dev()
    hm = _lib.Handle(device, precision, jit=getattr(m, "jit_kernels", True))
    opened.append(hm)
    m.stage_into(hm)
    return hm


def _by_model_shape(candidates, default):
    """Candidate indices grouped by the shape of the model they carry (first appearance order), or None
    when every candidate's model has the shape of `default` (the evaluator's own model, which candidates
    without a "model" entry run on): one plan on the evaluator's handle serves them all."""
    groups = {}
    for i, c in enumerate(candidates):
        m = c.get("model") if isinstance(c, dict) else None
        groups.setdefault(model_shape_key(default if m is None else m), []).append(i)
    if len(groups) == 1 and next(iter(groups)) == model_shape_key(default):
        return None
    return list(groups.values())


def _by_model(candidates, default):
    """Candidate indices grouped by the model OBJECT they carry (first appearance order)."""
    groups = {}
    for i, c in enumerate(candidates):
        m = c.get("model") if isinstance(c, dict) else None
        groups.setdefault(id(default if m is None else m), []).append(i)
    return list(groups.values())


def _needs_specialised_kernels(err):
    """The library's refusal of a model TABLE on a plan that runs the run-time-shape kernels (several models of one
    unregistered shape whose plugin is not built -- and, for one-evaluation models, never will be)."""
    return "several controller models in one plan" in str(err)


def _close_or_defer(evaluator, opened):
    """Close an evaluation's device objects (handles, plans) -- or, inside a model-shape group running next to
    other groups, hand them to the caller: freeing device memory synchronises the WHOLE device, so a group that
    finishes would stall the groups still running; the caller closes everything once all groups are done."""
    later = getattr(evaluator, "_close_later", None)
    if later is not None:
        later.extend(reversed(opened))
        return
    for obj in reversed(opened):
        obj.close()


def global_ids(index_offset, n):
    """Global indices of a shard's n candidates: index_offset is the first one's (a contiguous shard) or
    the array of all of them (a balanced shard, evaluate_sharded(..., weights=...))."""
    if np.ndim(index_offset) == 0:
        return int(index_offset) + np.arange(n, dtype=np.int64)
    ids = np.asarray(index_offset, dtype=np.int64).reshape(-1)
    if ids.shape[0] != n:
        raise ValueError("index_offset lists %d global indices for %d candidates" % (ids.shape[0], n))
    return ids


def _as_matrix(v, n):
    v = np.asarray(v, dtype=np.float64)
    return np.diag(v) if v.ndim == 1 else v.reshape(n, n)


def score_trajectories(cost, obs, ctrls):
    """``cost(traj)`` for a batch of trajectories obs [B,T+1,no], ctrls [B,T+1,nu]
    (Cost.__call__, cost.py:27-41): stage cost of every row plus the terminal cost of the last
    observation.  Quadratic costs are scored in closed form; anything else (threshold costs)
    falls back to the cost object's own per-step interface."""
    B = obs.shape[0]
    if is_quad_sum(cost):          # (also a sum of quadratics with different goals: costs/blocks.py)
        k = quad_sum_block(cost, obs.shape[2], ctrls.shape[2])
        d = obs - k["goal"]
        s = (np.einsum("bti,ij,btj->b", d, k["Q"], d) + np.einsum("bti,i->b", d, k["lin"])
             + obs.shape[1] * k["consts"][0] + np.einsum("bti,ij,btj->b", ctrls, k["R"], ctrls))
        return s + np.einsum("bi,ij,bj->b", d[:, -1], k["F"], d[:, -1]) + d[:, -1] @ k["lin_term"] + k["consts"][1]
    out = np.zeros(B)
    for b in range(B):
        for t in range(obs.shape[1]):
            out[b] += cost.eval_obs_cost(obs[b, t]) + cost.eval_ctrl_cost(ctrls[b, t])
        out[b] += cost.eval_term_obs_cost(obs[b, -1])
    return out


def _first_goal(cost, obs_dim):
    """Goal of the first quadratic term of a (nested) sum of quadratics (read structurally: the
    reference's SumCost.get_goal returns a cost object, sum_cost.py:45-47)."""
    c = cost
    while getattr(c, "costs", None) is not None and not callable(c.costs):
        c = c.costs[0]
    return np.asarray(c.get_goal(), dtype=np.float64).reshape(obs_dim).copy()


def candidate_cost_blocks(candidates, goal, obs_dim, ctrl_dim):
    """Stacked device cost blocks of a batch of candidates (Handle.set_cost_blocks) and their common
    ``terminal_goal`` flag.  A candidate gives its controller cost either as QuadCostFactory does
    (quad_cost_factory.py:64-95) -- ``Q``, ``R``, ``F`` diagonals or matrices about the task's goal --
    or as a cost OBJECT under ``"cost"``: a QuadCost or any sum of QuadCosts, shared goal or not,
    e.g. what ``QuadCostFactory + GaussRegFactory`` produce (gauss_reg_factory.py:37-45,
    sum_cost.py:49-54)."""
    no, nu = int(obs_dim), int(ctrl_dim)
    blocks = []
    for c in candidates:
        if c.get("cost") is not None:
            blocks.append(quad_sum_block(c["cost"], no, nu))
        else:
            blocks.append({"Q": _as_matrix(c["Q"], no), "R": _as_matrix(c["R"], nu), "F": _as_matrix(c["F"], no),
                           "goal": goal, "lin": np.zeros(no), "lin_term": np.zeros(no), "consts": np.zeros(2),
                           "terminal_goal": False})
    tg = {bool(b["terminal_goal"]) for b in blocks}
    if len(tg) > 1:
        raise TypeError("the candidates of one batch must agree on strict_reference")
    return stack_blocks(blocks), tg.pop()


def _task_goal(cost, obs_dim):
    """Goal the candidates' quadratic costs are centred on: the task cost's goal (as
    QuadCostFactory takes it, quad_cost_factory.py:64-66); for a sum without a shared goal the
    first term that has one; the origin if none has."""
    if is_quad_sum(cost):
        return _first_goal(cost, obs_dim)
    for c in [cost] + list(getattr(cost, "costs", [])):
        try:
            g = np.asarray(c.get_goal(), dtype=np.float64)
        except Exception:
            continue
        if g.shape == (obs_dim,):
            return g.copy()
    return np.zeros(obs_dim)


def _num_steps_lambda(num_steps):
    return lambda traj: len(traj) >= num_steps


def episode_of(task):
    """(max_steps, term_cond) of the episode ``PipelineTuner.eval_cfg`` simulates for `task`
    (tuning/pipeline_tuner.py:222-231): ``simulate(controller, init_obs, task.term_cond,
    max_steps=task.get_num_steps())`` when the task has a step count, simulate()'s own default
    of 10000 steps otherwise (utils/simulation.py:11).  term_cond is None when the task's
    termination condition is the one ``set_num_steps`` installs, ``len(traj) >= num_steps``
    (tasks/task.py:41-53) -- the episode length is then known up front, see
    ``default_episode_controls`` -- and the task's ``term_cond`` method otherwise (a user
    condition set with ``set_term_cond``; it has to be asked on the host)."""
    has_n = bool(task.has_num_steps())
    max_steps = int(task.get_num_steps()) if has_n else 10000
    probe = getattr(task, "has_user_term_cond", None)
    if probe is not None:                       # autompc_amd.Task
        user = bool(probe())
    else:                                       # a reference-style task object: recognise the
        fn = getattr(task, "_term_cond", None)  # closure set_num_steps installs by its code
        if fn is None:
            user = False
        else:
            ref = _num_steps_lambda(0)
            cells = [c.cell_contents for c in (getattr(fn, "__closure__", None) or ())]
            code = getattr(fn, "__code__", None)
            user = not (has_n and code is not None and code.co_code == ref.__code__.co_code
                        and code.co_consts == ref.__code__.co_consts and cells == [task.get_num_steps()])
    return max_steps, (task.term_cond if user else None)


def default_episode_controls(task):
    """Control steps of an episode that ends on ``len(traj) >= num_steps``: simulate() asks the
    condition after it has appended the new observation (utils/simulation.py:59-63), so the
    episode has num_steps rows = num_steps - 1 controls (one control when num_steps <= 1 -- the
    first question is asked about a two-row trajectory -- and none when max_steps = num_steps
    is 0).  Without a step count: simulate()'s default max_steps."""
    if not task.has_num_steps():
        return 10000
    n = int(task.get_num_steps())
    return min(n, max(1, n - 1))


class CandidateEvaluator:
    """Evaluates MPPI + QuadCost candidates for one (system, task, model, surrogate) on one GPU."""
    accepts_global_ids = True          # evaluate(index_offset=<array of global indices>) is understood

    def __init__(self, system, task, model, surrogate=None, precision="f64", device=0,
                 tile_rows=32, horizon_cap=30, term_check_every=8):
        """tile_rows / horizon_cap fix the rollout geometry (MppiPlan.set_geometry) so that a
        candidate's score is bit-identical for any batch it is evaluated in; horizon_cap defaults
        to the top of the reference's MPPI horizon range (mppi.py:52-55).  term_check_every: with
        a user termination condition the episode runs on the device in segments of this many
        control steps, the condition being asked on the host in between."""
        self.tile_rows, self.horizon_cap = int(tile_rows), int(horizon_cap)
        self.term_check_every = max(1, int(term_check_every))
        if not hasattr(model, "stage_into"):
            raise TypeError("needs a device-stageable model (autompc_amd.sysid.MLP)")
        self.system, self.task, self.model = system, task, model
        self.surrogate = surrogate if surrogate is not None else model
        # The device loop hands the surrogate's predicted STATE to the next solve.  That is
        # simulate()'s loop (controller.run -> model.update_state(state, u, obs), simulation.py:52-58)
        # exactly when the model state is the observation (MLP, SINDy), or when controller and
        # surrogate are one model whose update_state reproduces its own prediction (ARX).  A model
        # that REBUILDS its state from every observation (Koopman's lift, koopman.py:166-168) hands the
        # device its basis functions instead (device_lift): the loop then carries the surrogate's own
        # state and re-lifts the observation before every solve (ampc_mppi_plan_set_state_lift).
        self._lift = model.device_lift() if hasattr(model, "device_lift") else None
        if model.state_dim != system.obs_dim and self._lift is None:
            if not (getattr(model, "device_closed_loop", False) and self.surrogate is model):
                raise TypeError(
                    "%s keeps a model state that is not the observation and rebuilds it from every new "
                    "observation (update_state) in a way the device-resident closed loop cannot represent; "
                    "score such candidates with simulate() and the drop-in controller"
                    % type(model).__name__)
        # width of the state the loop carries and records
        self._snx = self.surrogate.state_dim if self._lift is not None else model.state_dim
        self.precision, self.device = precision, device
        b = task.get_ctrl_bounds()
        self.umin, self.umax = b[:, 0].copy(), b[:, 1].copy()
        self.goal = _task_goal(task.get_cost(), system.obs_dim)
        self.last_lengths = None

    def evaluate(self, candidates, n_steps=None, seed=0, init_obs=None, eps_all=None,
                 act_init=None, return_trajectories=False, index_offset=0, timing=None):
        """Closed-loop score of every candidate (a list of dicts with keys horizon, sigma, lmda,
        num_path, Q, R, F -- Q/R/F either diagonals or full matrices): the ``surr_cost`` of
        pipeline_tuner.py:213-258.

        Episode.  With ``n_steps=None`` (the tuner's call) the episode is the one eval_cfg
        simulates: ``task.term_cond`` with ``max_steps = task.get_num_steps()``, i.e.
        ``num_steps`` rows = ``num_steps - 1`` control steps for a task that only has a step
        count (``default_episode_controls``); a user termination condition (``set_term_cond``) is
        honoured exactly -- the device runs ``term_check_every`` steps at a time, the host asks the
        condition about every new row in order, a finished candidate leaves the batch and is
        scored on its own rows.  An explicit ``n_steps`` is ``simulate(..., max_steps=n_steps)``
        with no termination condition: exactly that many control steps.

        All randomness of candidate i is keyed by (seed, index_offset + i): its warm start comes
        from ``default_rng([seed, index_offset + i])`` and its device noise from the Philox
        stream of that global index.  A rank that evaluates the shard ``all[lo:hi]`` with
        ``index_offset=lo`` therefore returns exactly the scores a single process returns for
        those candidates.

        return_trajectories: also returns obs [B, rows, nx] and ctrls [B, rows, nu]; with a
        termination condition, candidates that stopped early are NaN-padded to the longest one
        and ``self.last_lengths`` holds every candidate's row count.

        timing: an optional dict; on return timing["timing"] holds the average HIP-event duration of
        the rollout / update kernels over the closed loop's control steps (bench.py roofline leg)
        and timing["control_steps"] the number of control steps the episode ran."""
        B = len(candidates)
        if B == 0:
            return (np.zeros(0), None, None) if return_trajectories else np.zeros(0)
        too_long = [int(c["horizon"]) for c in candidates if int(c["horizon"]) > self.horizon_cap]
        if too_long:
            # the LDS layout (and with it the summation order) would follow the batch's longest
            # horizon: scores would silently depend on what else is in the batch
            raise ValueError("candidate horizon %d exceeds the evaluator's horizon_cap %d; construct "
                             "CandidateEvaluator(horizon_cap=...) for the range being searched"
                             % (max(too_long), self.horizon_cap))
        ids = global_ids(index_offset, B)
        shape_groups = _by_model_shape(candidates, self.model)
        if shape_groups is not None:
            # Candidates that carry controller models of DIFFERENT shapes (pipeline.py:138-145: a model per
            # configuration): one plan per shape, evaluated in turn; same-shape models share a plan
            # (ampc_mppi_plan_set_models).  A candidate's randomness is keyed by its global index either way.
            if eps_all is not None or act_init is not None or timing is not None:
                raise ValueError("recorded noise / warm starts / kernel timing are laid out for ONE plan: evaluate "
                                 "the candidates of each model shape separately")
            return self._evaluate_shape_groups(shape_groups, candidates, ids, dict(
                n_steps=n_steps, seed=seed, init_obs=init_obs, return_trajectories=return_trajectories))
        opened = []                       # device objects, closed on every exit path
        try:
            return self._evaluate(candidates, n_steps, seed, init_obs, eps_all, act_init,
                                  return_trajectories, ids, opened, timing)
        except _lib.AmpcError as e:
            # several models of ONE shape share a plan through a model table, which needs the kernels specialised
            # for the shape; where those are not to be had (a shape that is not registered, on handles that do not
            # build plugins -- jit_kernels = False: models fitted for this one evaluation) every model gets a plan
            # of its own on the run-time-shape kernels.  Same scores: a candidate's randomness follows its index.
            by_model = _by_model(candidates, self.model)
            own_only = len(by_model) == 1 and (candidates[0].get("model") is None or candidates[0]["model"] is self.model)
            if not _needs_specialised_kernels(e) or own_only or eps_all is not None or act_init is not None \
                    or timing is not None:
                raise
            return self._evaluate_shape_groups(by_model, candidates, ids, dict(
                n_steps=n_steps, seed=seed, init_obs=init_obs, return_trajectories=return_trajectories))
        finally:
            _close_or_defer(self, opened)

    def _evaluate_shape_groups(self, groups, candidates, ids, kw):
        import copy
        B = len(candidates)
        scores, trajs, lengths = np.empty(B), [None] * B, np.zeros(B, dtype=np.int64)
        # every candidate's model is made explicit first: a candidate WITHOUT a "model" entry runs on the
        # evaluator's own model, whatever the first candidate of its shape group carries
        candidates = [dict(c, model=c.get("model") if c.get("model") is not None else self.model) for c in candidates]

        later = []                        # (list.extend is atomic under the GIL)

        def run_group(idx):
            sub = copy.copy(self)
            sub._close_later = later
            sub.model = candidates[idx[0]]["model"]       # the group's plan handle is staged with one of ITS models
            # (sub.surrogate stays the evaluator's simulation model: copy.copy kept the reference)
            return idx, sub, sub.evaluate([candidates[i] for i in idx], index_offset=ids[idx], **kw)
        # The groups are independent plans on handles (streams) of their own; most of a small group's time is host
        # work (staging the model, building the plan, freeing both), so they run side by side on a few host threads
        # -- the library calls release the GIL.  Results do not depend on the order (nothing is shared).
        from .hostpin import thread_cap
        workers = thread_cap(min(len(groups), int(getattr(self, "group_threads", 8))))
        try:
            if workers > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=workers) as pool:
                    results = list(pool.map(run_group, groups))
            else:
                results = [run_group(idx) for idx in groups]
        finally:
            for obj in later:
                obj.close()
        for idx, sub, out in results:
            if kw["return_trajectories"]:
                sc, ob, ct = out
                for k, i in enumerate(idx):
                    trajs[i] = (ob[k], ct[k])
            else:
                sc = out
            scores[idx] = sc
            if sub.last_lengths is not None:
                lengths[idx] = sub.last_lengths
        self.last_lengths = lengths if lengths.any() else None
        if not kw["return_trajectories"]:
            return scores
        L = max(t[0].shape[0] for t in trajs)
        if len({t[0].shape[1:] for t in trajs}) != 1:
            raise ValueError("return_trajectories: the shape groups record states of different widths; evaluate "
                             "them separately")
        obs = np.full((B, L) + trajs[0][0].shape[1:], np.nan)
        ctl = np.full((B, L) + trajs[0][1].shape[1:], np.nan)
        for i, (o, c) in enumerate(trajs):
            obs[i, :o.shape[0]], ctl[i, :c.shape[0]] = o, c
        return scores, obs, ctl

    # -- pieces ---------------------------------------------------------------------------------
    def _stage(self, candidates, opened):
        nu, no = self.system.ctrl_dim, self.system.obs_dim
        B = len(candidates)
        h = _lib.Handle(self.device, self.precision, jit=getattr(self.model, "jit_kernels", True))
        opened.append(h)
        self.model.stage_into(h)
        # controller models the candidates carry (same shape as the evaluator's: evaluate() groups by shape):
        # one handle each, holding only the weights; the plan gets their table (ampc_mppi_plan_set_models)
        models, self._model_index = candidate_models(candidates, self.model)
        self._model_handles = []
        if len(models) > 1 or models[0] is not self.model:
            self._model_handles = [model_handle(m, self.device, self.precision, opened) for m in models]
        blocks, _ = candidate_cost_blocks(candidates, self.goal, no, nu)
        h.set_cost_blocks(**blocks)
        h.set_ctrl_bounds(self.umin, self.umax)
        sur = None
        if self.surrogate is not self.model:
            sur = _lib.Handle(self.device, self.precision)
            opened.append(sur)
            self.surrogate.stage_into(sur)
        return h, sur

    def _plan(self, h, candidates, which, noise_ids, act_seq, opened):
        """MPPI plan over the candidates `which` (indices into the staged cost blocks)."""
        plan = _lib.MppiPlan(h, [int(candidates[i]["num_path"]) for i in which],
                             [int(candidates[i]["horizon"]) for i in which],
                             [float(candidates[i]["sigma"]) for i in which],
                             [float(candidates[i]["lmda"]) for i in which], cost_index=np.asarray(which))
        opened.append(plan)
        plan.set_geometry(self.tile_rows, self.horizon_cap)
        if self._lift is not None:
            plan.set_state_lift(*self._lift)
        plan.set_noise_ids(np.asarray(noise_ids)[which])
        if self._model_handles:
            plan.set_models(self._model_handles, self._model_index[np.asarray(which)])
        plan.upload(act_seq=act_seq)
        return plan

    def _evaluate(self, candidates, n_steps, seed, init_obs, eps_all, act_init, return_trajectories,
                  index_offset, opened, timing=None):
        nx, nu, no = self._snx, self.system.ctrl_dim, self.system.obs_dim
        B = len(candidates)
        if n_steps is not None:
            n_ctl, term_cond = int(n_steps), None
        else:
            max_steps, term_cond = episode_of(self.task)
            n_ctl = default_episode_controls(self.task) if term_cond is None else max_steps
        init_obs = self.task.get_init_obs() if init_obs is None else np.asarray(init_obs)
        h, sur = self._stage(candidates, opened)
        Hs = [int(c["horizon"]) for c in candidates]
        if act_init is None:
            # MPPI.__init__ / reset() draw the warm start ~ N(0, sigma) (mppi.py:97-99); here from
            # a stream seeded by (seed, global candidate index)
            act_init = np.concatenate([
                np.random.default_rng([int(seed), int(index_offset[i])]).normal(
                    scale=np.sqrt(c["sigma"]), size=Hs[i] * nu)
                for i, c in enumerate(candidates)])
        act_init = np.asarray(act_init, dtype=np.float64).ravel()
        noise_ids = index_offset               # (global candidate indices: global_ids)
        try:
            terms = cost_terms(self.task.get_cost(), no, nu)
        except TypeError:
            terms = None                  # a user-defined cost object: score it through its own
        if timing is not None:            # Python interface from the downloaded trajectories
            timing["control_steps"] = n_ctl
        init_obs = np.asarray(init_obs, dtype=np.float64)
        carried = self.surrogate if self._lift is not None else self.model
        if init_obs.shape == (nx,) and nx != no:
            state0 = init_obs
        else:                            # the state of the one-row trajectory simulate() starts from (simulation.py:44-47)
            from ..trajectory import Trajectory
            state0 = carried.traj_to_state(Trajectory(self.system, 1, init_obs[None, :no].copy(),
                                                      np.zeros((1, nu))))
        x0 = np.tile(state0, (B, 1))
        if term_cond is not None:
            return self._evaluate_segmented(candidates, h, sur, x0, n_ctl, term_cond, seed, eps_all,
                                            act_init, noise_ids, terms, return_trajectories, opened,
                                            timing)
        self.last_lengths = np.full(B, n_ctl + 1)
        if n_ctl == 0:                    # max_steps = 0: the one-row trajectory is scored as is
            obs, ctrls = x0[:, None, :].copy(), np.zeros((B, 1, nu))
            scores = self._score(h, terms, obs, ctrls)
            return (scores, obs, ctrls) if return_trajectories else scores
        plan = self._plan(h, candidates, np.arange(B), noise_ids, act_init, opened)
        if timing is not None:
            plan.set_timing(True)
        if terms is not None:
            res = plan.closed_loop_scored(x0, n_ctl, terms, seed=seed, eps_all=eps_all, surrogate=sur,
                                          return_trajectories=return_trajectories)
            scores, obs, ctrls = res if return_trajectories else (res, None, None)
        else:
            obs, ctrls = plan.closed_loop(x0, n_ctl, seed=seed, eps_all=eps_all, surrogate=sur)
            scores = self._score(h, None, obs, ctrls)
        if timing is not None:
            timing["timing"] = plan.timing()
        return (scores, obs, ctrls) if return_trajectories else scores

    def _score(self, h, terms, obs, ctrls):
        no = self.system.obs_dim
        if terms is not None:
            return h.score_trajectories(terms, obs, ctrls, obs_dim=no)
        return score_trajectories(self.task.get_cost(), obs[:, :, :no], ctrls)

    def _evaluate_segmented(self, candidates, h, sur, x0, max_steps, term_cond, seed, eps_all,
                            act_init, noise_ids, terms, return_trajectories, opened, timing):
        """An episode with a termination condition only the host can evaluate
        (utils/simulation.py:62-63).  A candidate's closed loop does not depend on what else is in
        the plan (fixed geometry, noise keyed by its global index and the step index), so running it
        a few steps past its end and dropping it from the batch at the next segment boundary leaves
        every row it keeps bit-identical to an unsegmented run."""
        from ..trajectory import Trajectory
        nx, nu, no = self._snx, self.system.ctrl_dim, self.system.obs_dim
        B = len(candidates)
        Hs = np.array([int(c["horizon"]) for c in candidates])
        a_off = np.concatenate([[0], np.cumsum(Hs * nu)])
        e_sz = np.array([int(c["num_path"]) * int(c["horizon"]) * nu for c in candidates])
        e_off = np.concatenate([[0], np.cumsum(e_sz)])
        if eps_all is not None:
            eps_all = np.asarray(eps_all, dtype=np.float64).reshape(-1, int(e_off[-1]))
            if eps_all.shape[0] < max_steps:
                raise ValueError("eps_all holds %d control steps, the episode may run %d"
                                 % (eps_all.shape[0], max_steps))
        obs = np.full((B, max_steps + 1, nx), np.nan)
        ctl = np.full((B, max_steps + 1, nu), np.nan)
        obs[:, 0] = x0
        lengths = np.full(B, max_steps + 1)
        alive = np.arange(B)
        acts = [act_init[a_off[i]:a_off[i + 1]] for i in range(B)]
        plan, plan_of, step0 = None, None, 0
        t_sum, t_cnt = {"rollout_ms": 0.0, "update_ms": 0.0}, 0
        while step0 < max_steps and alive.size:
            k = min(self.term_check_every, max_steps - step0)
            if plan is None:
                plan = self._plan(h, candidates, alive, noise_ids, np.concatenate([acts[i] for i in alive]),
                                  opened)
                plan_of = alive.copy()
                if timing is not None:
                    plan.set_timing(True)
            plan.set_step_offset(step0)
            eps_seg = None
            if eps_all is not None:
                eps_seg = np.concatenate([np.concatenate([eps_all[s, e_off[i]:e_off[i + 1]] for i in plan_of])
                                          for s in range(step0, step0 + k)])
            o, c = plan.closed_loop(obs[plan_of, step0], k, seed=seed, eps_all=eps_seg, surrogate=sur)
            obs[plan_of, step0 + 1:step0 + k + 1] = o[:, 1:]
            ctl[plan_of, step0:step0 + k] = c[:, :k]
            still = []
            for i in plan_of:
                done = False
                for t in range(step0 + 1, step0 + k + 1):      # rows 0..t exist, control row t is zero
                    rows_c = ctl[i, :t + 1].copy()
                    rows_c[t] = 0.0
                    if term_cond(Trajectory(self.system, t + 1, obs[i, :t + 1, :no].copy(), rows_c)):
                        lengths[i], done = t + 1, True
                        break
                if not done:
                    still.append(i)
            step0 += k
            if len(still) != len(plan_of):
                # the survivors continue in a smaller plan, warm starts carried over
                a, _, _, _ = plan.download(act_seq=True, u=False)
                off = np.concatenate([[0], np.cumsum(Hs[plan_of] * nu)])
                for j, i in enumerate(plan_of):
                    acts[i] = a[off[j]:off[j + 1]]
                if timing is not None:
                    tm = plan.timing()
                    for key in t_sum:
                        t_sum[key] += tm[key] * tm["count"]
                    t_cnt += tm["count"]
                plan.close()
                opened.remove(plan)
                plan = None
            alive = np.array(still, dtype=int)
        if timing is not None:
            if plan is not None:
                tm = plan.timing()
                for key in t_sum:
                    t_sum[key] += tm[key] * tm["count"]
                t_cnt += tm["count"]
            timing["timing"] = {"rollout_ms": t_sum["rollout_ms"] / max(t_cnt, 1),
                                "update_ms": t_sum["update_ms"] / max(t_cnt, 1), "count": t_cnt}
            timing["control_steps"] = int(step0)
        scores = np.empty(B)
        for L in np.unique(lengths):
            idx = np.nonzero(lengths == L)[0]
            o, c = obs[idx, :L].copy(), ctl[idx, :L].copy()
            c[:, L - 1] = 0.0                                  # simulate()'s trailing zero control row
            scores[idx] = self._score(h, terms, o, c)
            obs[idx, L:], ctl[idx, L - 1] = np.nan, 0.0
            ctl[idx, L:] = np.nan
        self.last_lengths = lengths
        if return_trajectories:
            Lmax = int(lengths.max())
            return scores, obs[:, :Lmax], ctl[:, :Lmax]
        return scores


class IlqrCandidateEvaluator:
    """Closed-loop surrogate scores of iLQR + QuadCost candidates -- the other controller the
    reference's tuner searches over (IterativeLQRFactory: horizon 5-25, control/ilqr.py:31-41;
    QuadCostFactory gains, quad_cost_factory.py:46-58) -- as ``PipelineTuner.eval_cfg`` computes
    them (tuning/pipeline_tuner.py:213-258): ``controller.reset()``, ``simulate(controller,
    init_obs, task.term_cond, sim_model=surrogate, max_steps=num_steps)``, ``cost(traj)``; a
    singular ``Quu`` (the reference's ``LinAlgError``, ilqr.py:179) scores ``inf`` (:236-239).

    ``IterativeLQR.run`` re-solves from a zero guess at every control step (ilqr.py:267-295 with the
    default ``reuse_feedback``).  Episodes of known length (the task only carries a step count) run
    entirely on the device (``ampc_ilqr_closed_loop_var``): ALL candidates, whatever their horizon,
    stream through the slots of one plan built for the longest horizon (a slot's sweep, line search
    and Jacobian refresh run over its own candidate's horizon), each slot carrying one candidate's
    episode -- solve, surrogate step, next solve -- without a host round trip and without waiting
    for the other slots' solves.  With a user termination condition (asked on the host after every
    step) a control step of the batch is one queue of device solves (``ampc_ilqr_solve_queue_var``)
    followed by one batched surrogate step.  Either way everything is deterministic and every solve
    is bit-identical to a one-problem solve of its own horizon: a candidate's score does not depend
    on the batch it is in or on the rank that evaluates it."""

    accepts_global_ids = True

    def __init__(self, system, task, model, surrogate=None, precision="f64", device=0, device_resident=True,
                 max_slots=1024, max_threads=32, one_plan=True):
        """device_resident: episodes of known length run entirely on the device (ampc_ilqr_closed_loop_var;
        a user termination condition is asked on the host, one queue of solves per control step);
        max_slots: problems solved side by side (one workgroup each);
        one_plan = False: the round-4 scheme, kept for comparison (tools/ilqr_eval_rate.py) -- one plan per
        horizon, the horizon groups evaluated concurrently on max_threads host threads (one plan and stream
        each); same scores bit for bit."""
        self.device_resident, self.max_slots, self.max_threads = bool(device_resident), int(max_slots), max(1, int(max_threads))
        self.one_plan = bool(one_plan)
        if not hasattr(model, "stage_into"):
            raise TypeError("needs a device-stageable model (autompc_amd.sysid.MLP)")
        if precision != "f64":
            raise ValueError("iLQR solves in f64 (control/ilqr.py: f32 is outside the parity mode)")
        self.system, self.task, self.model = system, task, model
        self.surrogate = surrogate if surrogate is not None else model
        # The loop hands the surrogate's predicted STATE to the next solve: simulate()'s loop exactly when
        # the model state is the observation (MLP, SINDy), or when controller and surrogate are ONE model
        # whose update_state reproduces its own prediction (ARX: the stacked history, arx.py:94-99,113-127;
        # up to 128 states, csrc/ilqr_wide.hpp).  Koopman re-lifts every observation (koopman.py:166-168):
        # score such candidates with simulate() and the drop-in controller.
        if model.state_dim != system.obs_dim and not (getattr(model, "device_closed_loop", False)
                                                     and self.surrogate is model):
            raise TypeError("IlqrCandidateEvaluator carries the model state from one solve to the next: the "
                            "observation itself (MLP, SINDy) or the state of a model that is its own surrogate "
                            "(ARX); score %s candidates with simulate() and the drop-in controller"
                            % type(model).__name__)
        self.precision, self.device = precision, device
        self.bounded = bool(task.are_ctrl_bounded())
        b = task.get_ctrl_bounds()
        self.umin, self.umax = b[:, 0].copy(), b[:, 1].copy()
        self.goal = _task_goal(task.get_cost(), system.obs_dim)
        self.last_lengths = None

    def evaluate(self, candidates, n_steps=None, seed=0, init_obs=None, return_trajectories=False,
                 index_offset=0, max_iter=50):
        """candidates: dicts with keys horizon, Q, R, F (diagonals or full matrices).  Episode as in
        CandidateEvaluator.evaluate (eval_cfg's, or exactly n_steps control steps).  seed and
        index_offset are accepted for interface compatibility (nothing here is random)."""
        B = len(candidates)
        if B == 0:
            return (np.zeros(0), None, None) if return_trajectories else np.zeros(0)
        shape_groups = _by_model_shape(candidates, self.model)
        if shape_groups is not None:       # controller models of different shapes: one plan per shape, in turn
            return CandidateEvaluator._evaluate_shape_groups(
                self, shape_groups, candidates, global_ids(index_offset, B),
                dict(n_steps=n_steps, init_obs=init_obs, return_trajectories=return_trajectories, max_iter=max_iter))
        opened = []
        try:
            return self._evaluate(candidates, n_steps, init_obs, return_trajectories, int(max_iter), opened)
        except _lib.AmpcError as e:           # (as CandidateEvaluator.evaluate: a plan per model where no table can be had)
            by_model = _by_model(candidates, self.model)
            own_only = len(by_model) == 1 and (candidates[0].get("model") is None or candidates[0]["model"] is self.model)
            if not _needs_specialised_kernels(e) or own_only:
                raise
            return CandidateEvaluator._evaluate_shape_groups(
                self, by_model, candidates, global_ids(index_offset, B),
                dict(n_steps=n_steps, init_obs=init_obs, return_trajectories=return_trajectories, max_iter=max_iter))
        finally:
            _close_or_defer(self, opened)

    def _evaluate(self, candidates, n_steps, init_obs, return_trajectories, max_iter, opened):
        from ..trajectory import Trajectory
        nx, nu, no = self.model.state_dim, self.system.ctrl_dim, self.system.obs_dim
        B = len(candidates)
        if n_steps is not None:
            n_ctl, term_cond = int(n_steps), None
        else:
            max_steps, term_cond = episode_of(self.task)
            n_ctl = default_episode_controls(self.task) if term_cond is None else max_steps
        init_obs = self.task.get_init_obs() if init_obs is None else np.asarray(init_obs)
        # one plan for every horizon (a slot runs over its own candidate's horizon, ampc_ilqr_*_var);
        # one_plan = False: one plan per horizon -- the problems of a plan share the horizon, not the cost
        groups = {}
        hz_all = np.array([int(c["horizon"]) for c in candidates], dtype=np.int32)
        for i, c in enumerate(candidates):
            groups.setdefault(int(hz_all.max()) if self.one_plan else int(c["horizon"]), []).append(i)
        sur = _lib.Handle(self.device, self.precision)
        opened.append(sur)
        self.surrogate.stage_into(sur)
        plans = {}
        device_loop = term_cond is None and n_ctl >= 1 and self.device_resident
        # controller models the candidates carry (one shape: evaluate() groups by shape): a handle each,
        # the plan gets their table (ampc_ilqr_plan_set_models), a queue problem / episode names its entry
        models, model_index = candidate_models(candidates, self.model)
        model_handles = []
        if len(models) > 1 or models[0] is not self.model:
            if not self.one_plan:
                raise ValueError("candidates that carry their own model need the one-plan evaluator (one_plan=True)")
            model_handles = [model_handle(m, self.device, self.precision, opened) for m in models]
        for H, idx in groups.items():
            h = _lib.Handle(self.device, self.precision, jit=getattr(self.model, "jit_kernels", True))
            opened.append(h)
            self.model.stage_into(h)
            blocks, term_goal = candidate_cost_blocks([candidates[i] for i in idx], self.goal, no, nu)
            h.set_cost_blocks(**blocks)
            if self.bounded:
                h.set_ctrl_bounds(self.umin, self.umax)
            slots = min(len(idx), self.max_slots) if (device_loop or self.one_plan) else len(idx)
            plan = _lib.IlqrPlan(h, slots, H, self.system.dt, cost_index=np.arange(slots),
                                 clip_to_bounds=self.bounded, terminal_goal=term_goal)
            opened.append(plan)
            if model_handles:
                plan.set_models(model_handles)
            plans[H] = (plan, np.array(idx))
        mi_all = model_index if model_handles else None
        obs = np.full((B, n_ctl + 1, nx), np.nan)
        ctl = np.full((B, n_ctl + 1, nu), np.nan)
        init_obs = np.asarray(init_obs, dtype=np.float64)
        if nx != no and init_obs.shape != (nx,):
            # the state of the one-row trajectory simulate() starts from (simulation.py:44-47)
            init_obs = self.model.traj_to_state(Trajectory(self.system, 1, init_obs[None, :no].copy(), np.zeros((1, nu))))
        obs[:, 0] = init_obs
        lengths = np.full(B, n_ctl + 1)
        failed = np.zeros(B, dtype=bool)          # singular Quu: the reference's LinAlgError -> inf
        alive = np.ones(B, dtype=bool)
        if device_loop:
            # The episode length is known up front: every candidate's whole episode runs on the device
            # (ampc_ilqr_closed_loop) -- solve, surrogate step, next solve without a host round trip, the
            # candidates of a horizon group streaming through the plan's slots.
            # Horizon groups are independent plans on handles (streams) of their own: they run side by
            # side, one host thread each (the library call releases the GIL) -- a group of three
            # candidates occupies three compute units, and the reference's horizon range 5..25 makes up
            # to 21 groups.
            self.last_iterations = np.zeros(B, dtype=np.int64)
            x0 = np.asarray(init_obs, dtype=np.float64)

            def run_group(item):
                plan, idx = item
                return idx, plan.closed_loop(np.tile(x0, (len(idx), 1)), n_ctl, cost_index=np.arange(len(idx)),
                                             max_iter=max_iter, surrogate=sur,
                                             horizon=hz_all[idx] if self.one_plan else None,
                                             model_index=None if mi_all is None else mi_all[idx])
            items = list(plans.values())
            if len(items) > 1:
                from concurrent.futures import ThreadPoolExecutor
                from .hostpin import thread_cap
                with ThreadPoolExecutor(max_workers=thread_cap(min(len(items), self.max_threads))) as pool:
                    results = list(pool.map(run_group, items))
            else:
                results = [run_group(items[0])]
            for idx, out in results:
                bad = out["failed"] != 0
                failed[idx[bad]] = True
                obs[idx], ctl[idx] = out["obs"], out["ctrls"]
                self.last_iterations[idx] = out["iterations"]
            n_ctl_host = 0
        else:
            n_ctl_host = n_ctl
        for t in range(n_ctl_host):
            if not alive.any():
                break
            for H, (plan, idx) in plans.items():
                live = alive[idx]
                if not live.any():
                    continue
                # (finished candidates keep their slot: the solve is per problem, their result is unused)
                x = np.where(live[:, None], obs[idx, t], obs[idx, 0])
                if self.one_plan:
                    out = plan.solve_queue(x, None, np.arange(len(idx)), max_iter=max_iter, gains=False,
                                           horizon=hz_all[idx], model_index=None if mi_all is None else mi_all[idx])
                else:
                    out = plan.solve(x, np.zeros((len(idx), H, nu)), max_iter=max_iter)
                u = out["ctrls"][:, 0]              # u = ubar_0 + K_0 (x - xbar_0) with x = xbar_0
                bad = live & (out["status"] == 1)
                failed[idx[bad]] = True
                alive[idx[bad]] = False
                go = live & ~bad
                if go.any():
                    nxt = sur.pred_batch(x[go], u[go])
                    sel = idx[go]
                    ctl[sel, t] = u[go]
                    obs[sel, t + 1] = nxt
            if term_cond is not None:
                for i in np.nonzero(alive)[0]:
                    rows_c = ctl[i, :t + 2].copy()
                    rows_c[t + 1] = 0.0
                    if term_cond(Trajectory(self.system, t + 2, obs[i, :t + 2, :no].copy(), rows_c)):
                        lengths[i] = t + 2
                        alive[i] = False
        try:
            terms = cost_terms(self.task.get_cost(), no, nu)
        except TypeError:
            terms = None
        scores = np.full(B, np.inf)
        ok = ~failed
        for L in np.unique(lengths[ok]):
            idx = np.nonzero(ok & (lengths == L))[0]
            o, c = obs[idx, :L].copy(), ctl[idx, :L].copy()
            c[:, L - 1] = 0.0                       # simulate()'s trailing zero control row
            if terms is not None:
                scores[idx] = sur.score_trajectories(terms, o, c, obs_dim=no)
            else:
                scores[idx] = score_trajectories(self.task.get_cost(), o[:, :, :no], c)
            obs[idx, L:], ctl[idx, L - 1] = np.nan, 0.0
            ctl[idx, L:] = np.nan
        self.last_lengths = lengths
        if return_trajectories:
            Lmax = int(lengths.max())
            return scores, obs[:, :Lmax], ctl[:, :Lmax]
        return scores


def random_ilqr_candidates(system, n, seed=0):
    """Candidates drawn from the reference's ranges: iLQR horizon 5-25 (control/ilqr.py:36-38),
    QuadCost diagonal gains log-uniform in [1e-3, 1e4] (quad_cost_factory.py:46-58)."""
    rng = np.random.default_rng(seed)
    no, nu = system.obs_dim, system.ctrl_dim
    return [dict(horizon=int(rng.integers(5, 26)), Q=10 ** rng.uniform(-3, 4, size=no),
                 R=10 ** rng.uniform(-3, 4, size=nu), F=10 ** rng.uniform(-3, 4, size=no)) for _ in range(n)]


def evaluate_sharded(local_eval, candidates, rank=None, world=None, device=None, stats=None, weights=None):
    """Score ``candidates`` with ``local_eval(sub_list, lo) -> scores`` on this rank's contiguous
    shard ``candidates[lo:hi]`` and all-gather the scores so every rank returns the full vector
    (candidate order).  ``lo`` is the global index of the shard's first candidate: an evaluator that
    keys its randomness by it (CandidateEvaluator.evaluate(..., index_offset=lo)) returns the same
    score for a candidate whatever the world size.

    weights: None -- contiguous shards (equal counts; fine for candidates in random order).  "auto" or
    one weight per candidate -- shards balanced by work (``candidate_work``: num_path x horizon;
    ``balanced_shards``), for batches that arrive sorted (a horizon sweep, an optimiser's ranked
    proposals): ``local_eval(sub_list, ids)`` then receives the ARRAY of the shard's global indices
    (``index_offset=ids`` is accepted by the evaluators), and the scores still come back in candidate
    order, the same for any world size.
    Uses the default torch.distributed process group when one is initialised; with none (or world
    size 1) it is a plain local evaluation.

    stats: an optional dict that receives what the exchange was -- "backend", "ranks_in_gather",
    "gather_ms" (wall time of the all-gather on this rank, synchronised), "device".

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
    if weights is None:
        spans = [shard_bounds(n, r, world) for r in range(world)]
        shards = [np.arange(a, b, dtype=np.int64) for a, b in spans]
        mine, key = candidates[spans[rank][0]:spans[rank][1]], spans[rank][0]
    else:
        w = [candidate_work(c) for c in candidates] if isinstance(weights, str) else weights
        if len(w) != n:
            raise ValueError("evaluate_sharded: %d weights for %d candidates" % (len(w), n))
        shards = balanced_shards(w, world)
        mine, key = [candidates[i] for i in shards[rank]], shards[rank]
        if len(key) and np.array_equal(key, np.arange(key[0], key[0] + len(key))):
            key = int(key[0])                            # (a shard that happens to be contiguous: its first index)
        if stats is not None:
            loads = [float(np.sum(np.asarray(w, dtype=np.float64)[ix])) for ix in shards]
            stats.update(shard_work=loads, heaviest_over_mean=max(loads) / max(np.mean(loads), 1e-300))
    n_mine = len(shards[rank])
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        # (a one-rank process group still takes the collective below: the same code runs at any N)
        if stats is not None:
            stats.update(backend=None, ranks_in_gather=1, gather_ms=0.0, device=None)
        return np.asarray(local_eval(mine, key), dtype=np.float64)
    error = None
    try:
        local = np.asarray(local_eval(mine, key), dtype=np.float64)
        if local.shape != (n_mine,):
            raise ValueError("local_eval returned %r scores for %d candidates" % (local.shape, n_mine))
    except Exception as e:           # noqa: BLE001 -- reported after the collective
        error, local = e, np.zeros(n_mine)
    per = max(len(ix) for ix in shards)                # equal-sized slots for the all-gather
    dev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl"
        else torch.device("cpu"))
    slot = torch.full((per + 1,), float("nan"), dtype=torch.float64, device=dev)
    slot[:n_mine] = torch.from_numpy(local).to(dev)
    slot[per] = 0.0 if error is None else 1.0          # failure marker of this rank
    gathered = torch.empty(world * (per + 1), dtype=torch.float64, device=dev)
    import time
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    dist.all_gather_into_tensor(gathered, slot)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    g = gathered.cpu().numpy().reshape(world, per + 1)
    if stats is not None:
        stats.update(backend=dist.get_backend(), ranks_in_gather=int(g.shape[0]),
                     gather_ms=1e3 * (time.perf_counter() - t0), device=str(dev))
    if error is not None:
        raise error
    failed = [r for r in range(world) if g[r, per] != 0.0]
    if failed:
        raise RuntimeError("candidate evaluation failed on rank(s) %s" % failed)
    out = np.empty(n)
    for r in range(world):
        out[shards[r]] = g[r, :len(shards[r])]
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
