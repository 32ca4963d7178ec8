"""The reference's pipeline ``Configuration`` <-> the evaluators' candidate dictionaries.

What it replaces.  The reference searches ONE joint ConfigSpace built by ``Pipeline.get_configuration_space``
(autompc/pipeline.py:90-105): the model factory's space under the prefix ``_model:``, the controller factory's
under ``_ctrlr:``, the cost factory's under ``_cost:``.  ``Pipeline.__call__`` (pipeline.py:107-168) splits a
configuration back into the three sub-configurations (``set_subspace_configuration``, utils/cs_utils.py:33-37:
every key that starts with the prefix, prefix stripped) and hands each to its factory:

    _ctrlr:horizon | sigma | lmda | num_path         MPPIFactory (control/mppi.py:48-64); iLQR: horizon only
                                                     (control/ilqr.py:36-41)
    _cost:<obs>_Q | <obs>_F | <ctrl>_R               QuadCostFactory (costs/quad_cost_factory.py:40-95): diagonal
                                                     gains by observation / control NAME; an absent key is gain 0
    _model:nonlintype | n_hidden_layers | hidden_size_1..4 | lr    MLPFactory (sysid/mlp.py:107-135)

``candidate_from_config`` performs that split for the batched evaluators (``CandidateEvaluator`` /
``IlqrCandidateEvaluator`` take dicts with horizon, sigma, lmda, num_path, Q, R, F and optionally ``model`` /
``model_cfg``); ``config_from_candidate`` is the inverse, so that what a tuner reports as ``inc_cfg`` is
something ``Pipeline.__call__`` accepts.  Configurations are read through ``get_dictionary()`` (ConfigSpace's
``Configuration``) or as plain mappings; ``DictConfiguration`` is the minimal object with that method, used
where ConfigSpace itself is not installed.  ``sample_pipeline_configs`` draws from the factories' ranges with
the reference's key names (what ``cs.sample_configuration(n)`` / SMAC's initial design would produce).
"""
import numpy as np

MODEL, CTRLR, COST = "_model", "_ctrlr", "_cost"
_MPPI_KEYS = ("sigma", "lmda", "num_path")


class DictConfiguration(dict):
    """A mapping that answers ``get_dictionary()`` like ConfigSpace's ``Configuration``."""

    def get_dictionary(self):
        return dict(self)


def config_dict(cfg):
    """The plain {name: value} view of a ConfigSpace ``Configuration`` or of a mapping."""
    if hasattr(cfg, "get_dictionary"):
        return dict(cfg.get_dictionary())
    return dict(cfg)


def subspace(d, prefix, delimiter=":"):
    """Entries of `d` under `prefix`, prefix stripped (cs_utils.set_subspace_configuration)."""
    pre = prefix + delimiter
    return {k[len(pre):]: v for k, v in d.items() if k[:len(pre)] == pre}


def candidate_from_config(system, cfg):
    """One evaluator candidate from a pipeline configuration; the configuration itself rides along under
    ``"cfg"`` (tuners report it back unchanged)."""
    d = config_dict(cfg)
    ctrl, cost, model = subspace(d, CTRLR), subspace(d, COST), subspace(d, MODEL)
    if "horizon" not in ctrl:
        raise KeyError("configuration has no %s:horizon (MPPIFactory / IterativeLQRFactory sub-space)" % CTRLR)
    cand = {"horizon": int(ctrl["horizon"])}
    if any(k in ctrl for k in _MPPI_KEYS):
        missing = [k for k in _MPPI_KEYS if k not in ctrl]
        if missing:
            raise KeyError("MPPI configuration lacks %s" % ", ".join("%s:%s" % (CTRLR, k) for k in missing))
        cand.update(sigma=float(ctrl["sigma"]), lmda=float(ctrl["lmda"]), num_path=int(ctrl["num_path"]))
    # QuadCostFactory.__call__ (quad_cost_factory.py:73-92): gains by name, absent -> 0
    cand["Q"] = np.array([float(cost.get("%s_Q" % n, 0.0)) for n in system.observations])
    cand["F"] = np.array([float(cost.get("%s_F" % n, 0.0)) for n in system.observations])
    cand["R"] = np.array([float(cost.get("%s_R" % n, 0.0)) for n in system.controls])
    if model:
        cand["model_cfg"] = dict(model)
    cand["cfg"] = cfg
    return cand


def candidates_from_configs(system, cfgs):
    return [candidate_from_config(system, c) for c in cfgs]


def _diagonal(v, n, name):
    v = np.asarray(v, dtype=np.float64)
    if v.ndim == 2:
        if np.any(v - np.diag(np.diag(v)) != 0.0):
            raise ValueError("%s is not diagonal: QuadCostFactory's configuration space holds diagonal gains only" % name)
        v = np.diag(v)
    if v.shape != (n,):
        raise ValueError("%s has shape %r, expected (%d,)" % (name, v.shape, n))
    return v


def config_from_candidate(system, cand):
    """The pipeline configuration (reference key names) of a candidate: the inverse of candidate_from_config.
    A candidate that came from a configuration returns that configuration."""
    if cand.get("cfg") is not None:
        return cand["cfg"]
    out = DictConfiguration()
    for k, v in (cand.get("model_cfg") or {}).items():
        out["%s:%s" % (MODEL, k)] = v
    out["%s:horizon" % CTRLR] = int(cand["horizon"])
    if "num_path" in cand:
        out["%s:sigma" % CTRLR] = float(cand["sigma"])
        out["%s:lmda" % CTRLR] = float(cand["lmda"])
        out["%s:num_path" % CTRLR] = int(cand["num_path"])
    no, nu = system.obs_dim, system.ctrl_dim
    for n, g in zip(system.observations, _diagonal(cand["Q"], no, "Q")):
        out["%s:%s_Q" % (COST, n)] = float(g)
    for n, g in zip(system.observations, _diagonal(cand["F"], no, "F")):
        out["%s:%s_F" % (COST, n)] = float(g)
    for n, g in zip(system.controls, _diagonal(cand["R"], nu, "R")):
        out["%s:%s_R" % (COST, n)] = float(g)
    return out


def sample_mlp_config(rng):
    """One draw from MLPFactory's space (mlp.py:107-135): activation, depth as the STRING the reference's
    categorical holds, the hidden sizes the depth activates (16-256), log-uniform learning rate 1e-5..1."""
    depth = int(rng.integers(1, 5))
    cfg = {"nonlintype": str(rng.choice(["relu", "tanh", "sigmoid", "selu"])), "n_hidden_layers": str(depth)}
    for i in range(depth):
        cfg["hidden_size_%d" % (i + 1)] = int(rng.integers(16, 257))
    cfg["lr"] = float(10 ** rng.uniform(-5, 0))
    return cfg


def sample_pipeline_configs(system, n, rng, controller="mppi", model_axis=False):
    """`n` configurations with the reference's key names, drawn from the factories' ranges: MPPI horizon 5-30,
    sigma 1e-4-2, lmda 0.1-2, num_path 100-1000 (mppi.py:48-64) or iLQR horizon 5-25 (ilqr.py:36-41); cost
    gains log-uniform in [1e-3, 1e4] (quad_cost_factory.py:40-61); with model_axis the MLP sub-space."""
    out = []
    for _ in range(int(n)):
        c = DictConfiguration()
        if model_axis:
            for k, v in sample_mlp_config(rng).items():
                c["%s:%s" % (MODEL, k)] = v
        if controller == "mppi":
            c["%s:horizon" % CTRLR] = int(rng.integers(5, 31))
            c["%s:sigma" % CTRLR] = float(rng.uniform(1e-4, 2.0))
            c["%s:lmda" % CTRLR] = float(rng.uniform(0.1, 2.0))
            c["%s:num_path" % CTRLR] = int(rng.integers(100, 1001))
        elif controller == "ilqr":
            c["%s:horizon" % CTRLR] = int(rng.integers(5, 26))
        else:
            raise ValueError("controller must be 'mppi' or 'ilqr'")
        for name in system.observations:
            c["%s:%s_Q" % (COST, name)] = float(10 ** rng.uniform(-3, 4))
        for name in system.observations:
            c["%s:%s_F" % (COST, name)] = float(10 ** rng.uniform(-3, 4))
        for name in system.controls:
            c["%s:%s_R" % (COST, name)] = float(10 ** rng.uniform(-3, 4))
        out.append(c)
    return out
