"""Batch ask/tell tuner over the device-resident candidate evaluator.

What it replaces.  ``PipelineTuner.run`` (reference autompc/tuning/pipeline_tuner.py:151-319)
hands SMAC one ``eval_cfg(cfg)`` at a time (:213-258) and afterwards walks SMAC's run history
into a ``PipelineTuneResult`` (:19-21, :273-313).  Here the search proposes candidates in
batches (``ask``), a whole batch is evaluated at once -- sharded over the ranks of the default
``torch.distributed`` group, one GPU each -- and reported back (``tell``); the same
``PipelineTuneResult`` fields come out, filled in evaluation order.  The proposal rule is
random search over the reference's configuration ranges (what SMAC's initial design is);
any other proposer can drive ``ask``/``tell`` by passing ``sampler``.

Both cost columns of the result are scores of the episode ``eval_cfg`` simulates --
``simulate(controller, init_obs, task.term_cond, max_steps=task.get_num_steps())`` after a
``controller.reset()`` (:222-231 surrogate, :244-251 true dynamics): ``num_steps`` rows =
``num_steps - 1`` control steps unless the task carries a user termination condition, which is
honoured on both paths (CandidateEvaluator.evaluate).

Scores that are not finite (a diverged rollout) count as ``inf``, as ``eval_cfg`` does for a
controller that raises ``LinAlgError`` (:236-239).
"""
import time
from collections import namedtuple

import numpy as np

from .batch_eval import (IlqrCandidateEvaluator, evaluate_sharded, global_ids, random_candidates,
                         random_ilqr_candidates)
from .configs import (DictConfiguration, candidates_from_configs, config_from_candidate, sample_mlp_config)

# same fields, same order as the reference's namedtuple (pipeline_tuner.py:19-21)
PipelineTuneResult = namedtuple("PipelineTuneResult", [
    "inc_cfg", "cfgs", "inc_cfgs", "costs", "inc_costs", "truedyn_costs", "inc_truedyn_costs",
    "surr_trajs", "truedyn_trajs", "surr_tune_result"])


class BatchPipelineTuner:
    """``evaluator.evaluate(candidates, seed=..., index_offset=...) -> scores`` is the only thing
    required of the evaluator (autompc_amd.tuning.CandidateEvaluator provides it).  ``index_offset`` is the
    global evaluation index of the shard's first candidate -- an ``int`` -- unless the evaluator sets the class
    attribute ``accepts_global_ids = True`` (both in-tree evaluators do): shards are then balanced by work and
    ``index_offset`` may be the ARRAY of the shard's global indices (``tuning.batch_eval.global_ids`` turns either
    form into indices).  ``balance=True`` / ``False`` overrides the choice."""

    def __init__(self, system, evaluator, batch_size=64, sampler=None, truedyn_noise="device",
                 eval_kwargs=None, keep_trajs=False, balance=None, models=None, model_factory=None,
                 trajs=None, as_configs=False):
        """truedyn_noise: the noise mode of the controllers scored against the true dynamics
        (MPPI(noise=...): "device" Philox, or "numpy" / "numpy_device" = the reference's global
        legacy stream).  eval_kwargs: extra keyword arguments for every ``evaluator.evaluate`` call
        (e.g. a recorded noise stream to replay).

        model_factory + trajs: the model axis as the reference runs it -- ``eval_cfg`` calls
        ``pipeline(cfg, task, trajs)``, which instantiates AND TRAINS the configuration's model
        (pipeline.py:138-145 -> ModelFactory.__call__, sysid/model.py:24-48).  Candidates then carry a
        ``model_cfg`` (the `_model:` sub-configuration; the default sampler draws it from MLPFactory's
        ranges); before a shard is evaluated its models are built with ``model_factory(cfg, trajs,
        skip_train_model=True)`` and fitted together by ``sysid.mlp_fit.fit_mlps`` (lockstep PyTorch-ROCm fit,
        HIP-graph captured; each model exactly as its own ``train(trajs)``), their parameters staged from
        device memory (ampc_set_mlp_dev).  Fits are cached by configuration; every rank fits only the models
        of its own shard.  ``fit_seconds`` / ``eval_seconds`` accumulate what the two phases took.

        as_configs: report ``cfgs`` / ``inc_cfg`` as pipeline configurations with the reference's key names
        (`_ctrlr:horizon`, `_cost:<obs>_Q`, `_model:lr`, ...; tuning/configs.py) instead of candidate dicts;
        candidates that came from configurations (``run(..., configs=...)``) are always reported as those."""
        self.system, self.evaluator = system, evaluator
        self.truedyn_noise = truedyn_noise
        self.eval_kwargs = dict(eval_kwargs or {})
        # keep_trajs: also record every candidate's surrogate trajectory as (obs rows, control rows)
        # lists, what the reference keeps in info["surr_traj"] (pipeline_tuner.py:234) and returns
        # as PipelineTuneResult.surr_trajs; they travel between ranks with all_gather_object
        self.keep_trajs = bool(keep_trajs)
        # balance: the shards of a batch are balanced by work (num_path x horizon; evaluate_sharded's
        # weights="auto") instead of being contiguous; scores are the same either way
        # (None: balanced when the evaluator declares that it takes an ARRAY of global indices as
        # index_offset -- the in-tree evaluators do; an evaluator written against the documented integer
        # interface gets contiguous shards)
        self.balance = bool(getattr(evaluator, "accepts_global_ids", False)) if balance is None else bool(balance)
        # models: the model axis of the search -- the reference's pipeline configuration space joins
        # `_model:`, `_ctrlr:` and `_cost:` sub-spaces (pipeline.py:90-105) and eval_cfg builds (trains) the
        # model of every configuration (pipeline.py:138-145).  Here the candidate models are given
        # (trained beforehand, on the host or with torch on the GPU: sysid stays PyTorch); the default
        # sampler draws each candidate's "model" among them, the evaluators run candidates with
        # different models in one batch (ampc_*_plan_set_models).  Custom samplers may set c["model"] too.
        self.models = list(models) if models else None
        self.model_factory, self.trajs = model_factory, trajs
        if model_factory is not None and trajs is None:
            raise ValueError("model_factory needs the training trajectories (trajs=...)")
        self.as_configs = bool(as_configs)
        self._fitted = {}                      # model configuration -> fitted model
        self.fit_seconds = self.eval_seconds = 0.0
        self.models_fitted = 0
        self.batch_size = int(batch_size)
        if self.batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        self._sampler = sampler if sampler is not None else self._random_search
        self.reset()

    def reset(self):
        self.cfgs, self.costs, self.inc_cfgs, self.inc_costs = [], [], [], []
        self.truedyn_costs, self.inc_truedyn_costs = [], []
        self.surr_trajs = []
        self._inc_cfg, self._inc_cost, self._inc_truedyn = None, float("inf"), None

    def _random_search(self, n, rng):
        draw = random_ilqr_candidates if isinstance(self.evaluator, IlqrCandidateEvaluator) else random_candidates
        cands = draw(self.system, n, seed=int(rng.integers(1 << 31)))
        if self.models:
            for c, k in zip(cands, rng.integers(len(self.models), size=n)):
                c["model"], c["model_index"] = self.models[int(k)], int(k)
        elif self.model_factory is not None:
            draw_cfg = getattr(self.model_factory, "sample_configuration", None)
            for c in cands:
                c["model_cfg"] = dict(draw_cfg(rng)) if draw_cfg else sample_mlp_config(rng)
        return cands

    # -- the model axis: fit what a shard asks for (pipeline.py:138-145) ---------------------------------
    @staticmethod
    def _cfg_key(cfg):
        return tuple(sorted((str(k), repr(v)) for k, v in cfg.items()))

    def fit_models(self, candidates):
        """Give every candidate that carries a ``model_cfg`` (and no ``model`` yet) its trained model."""
        import time
        want = [c for c in candidates if isinstance(c, dict) and c.get("model") is None and c.get("model_cfg")]
        if not want:
            return
        if self.model_factory is None:
            raise ValueError("candidates carry a model configuration (`_model:` keys) but the tuner has no "
                             "model_factory / trajs to build and fit them with")
        t0 = time.perf_counter()
        fresh = {}
        for c in want:
            key = self._cfg_key(c["model_cfg"])
            if key not in self._fitted and key not in fresh:
                fresh[key] = self.model_factory(DictConfiguration(c["model_cfg"]), self.trajs, skip_train_model=True)
                # a model per configuration lives for its candidates' evaluation: no 5 s kernel build for it
                # (a plugin already cached for its shape is still used; the incumbent's controller, built by
                #  Pipeline.__call__ afterwards, is a new model object with the default)
                fresh[key].jit_kernels = False
        if fresh:
            from ..sysid.mlp import MLP
            from ..sysid.mlp_fit import fit_mlps
            mlps = [m for m in fresh.values() if isinstance(m, MLP)]
            if mlps:
                fit_mlps(mlps, self.trajs)
            for m in fresh.values():
                if not isinstance(m, MLP):
                    m.train(self.trajs, silent=True)
            self._fitted.update(fresh)
            self.models_fitted += len(fresh)
        for c in want:
            c["model"] = self._fitted[self._cfg_key(c["model_cfg"])]
        self.fit_seconds += time.perf_counter() - t0

    # -- ask / tell ---------------------------------------------------------------------------
    def ask(self, n, rng):
        """The next `n` candidates to evaluate."""
        cands = list(self._sampler(int(n), rng))
        if len(cands) != n:
            raise ValueError("sampler returned %d candidates, %d asked" % (len(cands), n))
        return cands

    def tell(self, candidates, scores, truedyn_scores=None):
        """Record evaluated candidates in order; keeps the incumbent trace the reference builds
        from SMAC's run history (pipeline_tuner.py:279-291: strict improvement replaces).
        truedyn_scores (optional) are recorded next to the surrogate scores and never steer the
        search (pipeline_tuner.py:181-184)."""
        scores = np.asarray(scores, dtype=np.float64)
        if len(candidates) != scores.shape[0]:
            raise ValueError("one score per candidate expected")
        if truedyn_scores is not None and len(truedyn_scores) != len(candidates):
            raise ValueError("one true-dynamics score per candidate expected")
        for i, (cfg, s) in enumerate(zip(candidates, scores)):
            s = float(s) if np.isfinite(s) else float("inf")
            td = None if truedyn_scores is None else float(truedyn_scores[i])
            if isinstance(cfg, dict) and (self.as_configs or cfg.get("cfg") is not None):
                cfg = config_from_candidate(self.system, cfg)
            if s < self._inc_cost or self._inc_cfg is None:
                self._inc_cost, self._inc_cfg, self._inc_truedyn = s, cfg, td
            self.cfgs.append(cfg)
            self.costs.append(s)
            self.inc_cfgs.append(self._inc_cfg)
            self.inc_costs.append(self._inc_cost)
            if td is not None:
                self.truedyn_costs.append(td)
                self.inc_truedyn_costs.append(self._inc_truedyn)

    def result(self):
        return PipelineTuneResult(inc_cfg=self._inc_cfg, cfgs=list(self.cfgs),
                                  inc_cfgs=list(self.inc_cfgs), costs=list(self.costs),
                                  inc_costs=list(self.inc_costs),
                                  truedyn_costs=list(self.truedyn_costs),
                                  inc_truedyn_costs=list(self.inc_truedyn_costs),
                                  surr_trajs=list(self.surr_trajs),
                                  truedyn_trajs=[], surr_tune_result=None)

    # -- the loop -----------------------------------------------------------------------------
    def truedyn_score(self, cand, truedyn, seed=0):
        """Score of one candidate's controller against the true dynamics ``truedyn(obs, ctrl) ->
        obs`` (eval_cfg's second branch, pipeline_tuner.py:241-256): the MPPI solves run on the
        device, the dynamics callback runs on the host between them."""
        from .. import MPPI, IterativeLQR, QuadCost, Task, simulate
        ev = self.evaluator
        no, nu = self.system.obs_dim, self.system.ctrl_dim

        def mat(v, n):
            v = np.asarray(v, dtype=np.float64)
            return np.diag(v) if v.ndim == 1 else v.reshape(n, n)
        task = Task(self.system)
        task.set_cost(QuadCost(self.system, mat(cand["Q"], no), mat(cand["R"], nu), mat(cand["F"], no),
                               goal=ev.goal))
        task.set_ctrl_bounds(ev.umin, ev.umax)
        if "num_path" not in cand:            # an iLQR candidate (horizon + cost weights)
            ctl = IterativeLQR(self.system, task, cand.get("model") or ev.model, int(cand["horizon"]),
                               precision=ev.precision, device=ev.device)
        else:
            ctl = MPPI(self.system, task, cand.get("model") or ev.model, horizon=int(cand["horizon"]),
                       num_path=int(cand["num_path"]), sigma=float(cand["sigma"]),
                       lmda=float(cand["lmda"]), noise=self.truedyn_noise, seed=seed,
                       precision=ev.precision, device=ev.device)
        ctl.reset()
        kw = {"max_steps": ev.task.get_num_steps()} if ev.task.has_num_steps() else {}
        traj = simulate(ctl, ev.task.get_init_obs(), ev.task.term_cond, dynamics=truedyn, **kw)
        return float(ev.task.get_cost()(traj))

    @staticmethod
    def _gather_trajs(kept, n):
        """Every rank's {batch index: trajectory} -> the batch's trajectories in candidate order."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            parts = [None] * dist.get_world_size()
            dist.all_gather_object(parts, kept)
            kept = {k: v for part in parts for k, v in part.items()}
        return [kept[i] for i in range(n)]

    def run(self, n_iters, rng, seed=0, truedyn=None, configs=None):
        """Evaluate `n_iters` candidates in batches of `batch_size`.  configs: pipeline configurations
        (ConfigSpace ``Configuration`` objects or mappings with the reference's `_model:` / `_ctrlr:` /
        `_cost:` keys, e.g. ``cs.sample_configuration(512)`` or SMAC's initial design) to evaluate INSTEAD of
        sampling -- the first `n_iters` of them, in order; they come back as ``cfgs`` / ``inc_cfg``.  Every rank must call this
        with an identically seeded `rng` (proposals are drawn redundantly on every rank so that
        no broadcast is needed); each rank evaluates its contiguous shard of every batch and the
        scores are all-gathered.  With `truedyn`, every candidate is also scored against the true
        dynamics (recorded, not used by the search).  Returns (incumbent, PipelineTuneResult)."""
        done = 0
        while done < n_iters:
            n = min(self.batch_size, n_iters - done)
            if configs is not None:
                if len(configs) < done + n:
                    raise ValueError("%d configurations given, %d evaluations asked" % (len(configs), n_iters))
                batch = candidates_from_configs(self.system, configs[done:done + n])
            else:
                batch = self.ask(n, rng)
            # randomness keyed by (seed, global evaluation index): scores do not depend on the
            # world size or on the batch size
            kept = {}

            def local(shard, lo, d=done):
                # (lo: the shard's first index in the batch, or -- balanced shards -- all of its indices)
                self.fit_models(shard)
                t0 = time.perf_counter()
                try:
                    return evaluate(shard, lo, d)
                finally:
                    self.eval_seconds += time.perf_counter() - t0

            def evaluate(shard, lo, d):
                if not self.keep_trajs:
                    return self.evaluator.evaluate(shard, seed=seed, index_offset=d + lo, **self.eval_kwargs)
                sc, obs, ctl = self.evaluator.evaluate(shard, seed=seed, index_offset=d + lo,
                                                       return_trajectories=True, **self.eval_kwargs)
                lens = getattr(self.evaluator, "last_lengths", None)
                no = self.system.obs_dim
                ids = global_ids(lo, len(shard))
                for i in range(len(shard)):
                    L = int(lens[i]) if lens is not None else obs.shape[1]
                    kept[int(ids[i])] = (obs[i, :L, :no].tolist(), ctl[i, :L].tolist())
                return sc
            # shards balanced by work (num_path x horizon): a batch an optimiser hands over sorted would
            # otherwise load the ranks unevenly
            scores = evaluate_sharded(local, batch, weights="auto" if self.balance else None)
            if self.keep_trajs:
                self.surr_trajs.extend(self._gather_trajs(kept, n))
            td = None
            if truedyn is not None:
                td = evaluate_sharded(
                    lambda shard, lo, d=done: [self.truedyn_score(c, truedyn, seed=seed + d + lo + i)
                                               for i, c in enumerate(shard)], batch)
            self.tell(batch, scores, td)
            done += n
        return self._inc_cfg, self.result()
