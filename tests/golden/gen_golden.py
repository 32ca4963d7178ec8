#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (williamedwards/autompc).

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden.py

The reference is imported unmodified from /root/reference with four absent
third-party packages stubbed (ConfigSpace, smac, pysindy, gpytorch) and
scipy.linalg.pinv2 aliased.  None of the stubbed packages is on
the MPPI / iLQR / MLP / QuadCost path.  Only DATA is written: ``*.npz`` files
with inputs (or the seeds that regenerate them) and the reference's outputs.

MLP weights are not stored: they are regenerated from ``oracle.mlp.random_params
(seed)`` (numpy PCG64) on both sides and loaded into the reference's torch net
with ``load_state_dict``; a weight checksum is stored to detect RNG drift.
MPPI noise comes from the legacy global ``np.random.seed`` stream, exactly as
the reference draws it.
"""
import contextlib
import io
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def install_reference():
    import scipy.linalg
    names = ["ConfigSpace", "ConfigSpace.hyperparameters", "ConfigSpace.conditions",
             "ConfigSpace.forbidden", "smac", "smac.scenario", "smac.scenario.scenario",
             "smac.facade", "smac.facade.smac_hpo_facade", "pysindy", "pysindy.differentiation",
             "gpytorch", "gpytorch.models", "gpytorch.variational", "gpytorch.mlls",
             "gpytorch.distributions", "gpytorch.kernels", "gpytorch.means", "gpytorch.likelihoods"]
    for n in names:
        sys.modules[n] = MagicMock()
    sys.modules["pysindy.differentiation"].base.BaseDifferentiation = type("BaseDifferentiation", (), {})
    sys.modules["pysindy"].differentiation = sys.modules["pysindy.differentiation"]
    for cls in ("ExactGP", "ApproximateGP"):
        t = type(cls, (), {})
        setattr(sys.modules["gpytorch"].models, cls, t)
        setattr(sys.modules["gpytorch.models"], cls, t)
    if not hasattr(scipy.linalg, "pinv2"):
        scipy.linalg.pinv2 = scipy.linalg.pinv
    sys.path.insert(0, "/root/reference")


install_reference()
with contextlib.redirect_stdout(io.StringIO()):
    import autompc as ampc                                   # noqa: E402
    from autompc.control.mppi import MPPI                    # noqa: E402
    from autompc.control.ilqr import IterativeLQR            # noqa: E402
    from autompc.sysid.mlp import MLP                        # noqa: E402
    from autompc.sysid.model import Model                    # noqa: E402
    from autompc.costs import QuadCost                       # noqa: E402
    from autompc.tasks import Task                           # noqa: E402
    from autompc.utils.simulation import simulate            # noqa: E402
import torch                                                 # noqa: E402

from oracle import mlp as omlp                               # noqa: E402
from oracle.analytic import CubicIntegrator                  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_system(nx, nu, dt=0.05):
    return ampc.System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=dt)


def normalisers(nx, nu, seed):
    rng = np.random.default_rng(seed + 7919)
    return (rng.normal(scale=0.3, size=nx + nu), rng.uniform(0.5, 2.0, size=nx + nu),
            rng.normal(scale=0.02, size=nx), rng.uniform(0.05, 0.2, size=nx))


def ref_mlp(system, hidden, activation, seed, plain_norm=False):
    """Reference MLP carrying the deterministic numpy weights."""
    nx, nu = system.obs_dim, system.ctrl_dim
    p = omlp.random_params(nx, nu, hidden, activation, seed=seed)
    if not plain_norm:
        p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"] = normalisers(nx, nu, seed)
    kw = {"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)}
    model = quiet(MLP, system, n_hidden_layers=len(hidden), nonlintype=activation,
                  use_cuda=False, **kw)
    sd = {}
    for i in range(len(hidden)):
        sd["layers.layer%d.weight" % i] = torch.from_numpy(p["weights"][i])
        sd["layers.layer%d.bias" % i] = torch.from_numpy(p["biases"][i])
    sd["output_layer.weight"] = torch.from_numpy(p["weights"][-1])
    sd["output_layer.bias"] = torch.from_numpy(p["biases"][-1])
    model.net.load_state_dict(sd)
    model.xu_means, model.xu_std = p["xu_means"], p["xu_std"]
    model.dy_means, model.dy_std = p["dy_means"], p["dy_std"]
    return model, p


def weight_checksum(p):
    return np.array([sum(float(np.sum(w)) for w in p["weights"]),
                     sum(float(np.sum(np.abs(b))) for b in p["biases"]),
                     float(p["weights"][0][0, 0]), float(p["weights"][-1][-1, -1])])


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-40s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- MLP
MLP_CASES = [
    # tag, nx, nu, hidden, activation, seed
    ("p64_relu", 2, 1, [64, 64], "relu", 11),
    ("p64_tanh", 2, 1, [64, 64], "tanh", 12),
    ("hc1_relu", 17, 1, [256, 256], "relu", 13),
    ("hc6_relu", 17, 6, [256, 256], "relu", 14),
    ("hc6_tanh", 17, 6, [256, 256], "tanh", 15),
    ("hc6_sigmoid", 17, 6, [256, 256], "sigmoid", 16),
    ("hc6_selu", 17, 6, [256, 256], "selu", 17),
    ("cp3_selu", 4, 1, [32, 16, 8], "selu", 18),
    ("odd1_sigmoid", 4, 2, [100], "sigmoid", 19),
    ("deep4_tanh", 3, 2, [48, 200, 17, 64], "tanh", 20),
]


def gen_mlp():
    for tag, nx, nu, hidden, act, seed in MLP_CASES:
        system = make_system(nx, nu)
        model, p = ref_mlp(system, hidden, act, seed)
        rng = np.random.default_rng(seed + 1000)
        states = rng.normal(size=(32, nx))
        ctrls = rng.normal(size=(32, nu))
        pb = model.pred_batch(states, ctrls)
        db, jx, ju = model.pred_diff_batch(states, ctrls)
        p0 = model.pred(states[0], ctrls[0])
        d0, jx0, ju0 = model.pred_diff(states[0], ctrls[0])
        save("mlp_" + tag, nx=nx, nu=nu, hidden=np.array(hidden), activation=act, seed=seed,
             wsum=weight_checksum(p), states=states, ctrls=ctrls, pred_batch=pb,
             diff_pred=db, diff_jx=jx, diff_ju=ju, pred0=p0, diff0_pred=d0, diff0_jx=jx0,
             diff0_ju=ju0)


# -------------------------------------------------------------------------- cost
def gen_cost():
    system = make_system(5, 3)
    rng = np.random.default_rng(5)
    Q, R, F = rng.normal(size=(5, 5)), rng.normal(size=(3, 3)), rng.normal(size=(5, 5))
    goal = rng.normal(size=5)
    cost = QuadCost(system, Q, R, F, goal=goal)
    obs, ctrl = rng.normal(size=5), rng.normal(size=3)
    o = cost.eval_obs_cost_hess(obs)
    c = cost.eval_ctrl_cost_hess(ctrl)
    t = cost.eval_term_obs_cost_hess(obs)
    od = cost.eval_obs_cost_diff(obs)
    td = cost.eval_term_obs_cost_diff(obs)
    # trajectory score (Cost.__call__)
    traj = ampc.zeros(system, 7)
    traj.obs[:] = rng.normal(size=(7, 5))
    traj.ctrls[:] = rng.normal(size=(7, 3))
    # the reference's own known answers (tests/test_costs.py:192-205): SumCost of
    # QuadCost(Q=I,R=I,F=I) + QuadCost(Q=diag(1,5), ...) evaluated at obs [-1, 1]
    save("cost_quad", Q=Q, R=R, F=F, goal=goal, obs=obs, ctrl=ctrl,
         obs_cost=cost.eval_obs_cost(obs), ctrl_cost=cost.eval_ctrl_cost(ctrl),
         term_cost=cost.eval_term_obs_cost(obs),
         obs_c=o[0], obs_j=o[1], obs_h=o[2], ctrl_c=c[0], ctrl_j=c[1], ctrl_h=c[2],
         term_c=t[0], term_j=t[1], term_h=t[2], obs_diff_j=od[1], term_diff_c=td[0],
         term_diff_j=td[1], traj_obs=traj.obs, traj_ctrls=traj.ctrls, traj_cost=cost(traj))


# -------------------------------------------------------------------------- MPPI
MPPI_CASES = [
    # tag, nx, hidden, act, mlpseed, N, H, sigma, lmda, bounds, costkind, npseed
    ("c2_pendulum", 2, [64, 64], "relu", 31, 1024, 30, 1.0, 1.0, (-2.0, 2.0), "plain", 0),
    ("hc_nu1", 17, [256, 256], "relu", 32, 256, 20, 1.0, 1.0, (-1.0, 1.0), "plain", 1),
    ("clip_asym", 2, [64, 64], "tanh", 33, 200, 12, 1.5, 0.7, (-0.3, 0.5), "plain", 2),
    ("lowlmda_goal", 4, [32, 32], "tanh", 34, 300, 15, 0.5, 0.1, (-1.5, 1.5), "dense", 3),
]


def make_cost(system, kind, seed):
    no, nu = system.obs_dim, system.ctrl_dim
    if kind == "plain":
        return QuadCost(system, np.eye(no), 0.01 * np.eye(nu), np.eye(no), goal=np.zeros(no))
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(no, no))
    B = rng.normal(size=(no, no))
    Q = A @ A.T / no + 0.1 * rng.normal(size=(no, no))      # deliberately non-symmetric
    F = B @ B.T / no
    R = np.diag(rng.uniform(0.01, 0.1, size=nu)) + 0.005
    goal = rng.normal(scale=0.2, size=no)
    return QuadCost(system, Q, R, F, goal=goal)


def gen_mppi():
    for tag, nx, hidden, act, mseed, N, H, sigma, lmda, bnd, ckind, npseed in MPPI_CASES:
        system = make_system(nx, 1)
        model, p = ref_mlp(system, hidden, act, mseed, plain_norm=(ckind == "plain"))
        task = Task(system)
        cost = make_cost(system, ckind, mseed)
        task.set_cost(cost)
        task.set_ctrl_bound("u0", bnd[0], bnd[1])
        Q, R, F = cost.get_cost_matrices()
        np.random.seed(npseed)
        ctl = quiet(MPPI, system, task, model, horizon=H, num_path=N, sigma=sigma, lmda=lmda)
        out = {"act0": ctl.act_sequence.copy()}
        obs = np.random.default_rng(npseed + 99).uniform(-0.1, 0.1, size=nx)
        constate = np.concatenate([obs, np.zeros(1)])
        n_runs = 4
        for r in range(n_runs):
            if r == 3:
                quiet(ctl.reset)
                out["act_reset"] = ctl.act_sequence.copy()
            # capture costs/eps of this solve by wrapping update()
            cap = {}
            orig_update = ctl.update

            def spy(costs, eps, cap=cap, orig=orig_update):
                cap["costs"] = costs.copy()
                cap["eps"] = eps.copy()
                return orig(costs, eps)
            ctl.update = spy
            u, constate = ctl.run(constate, obs)
            ctl.update = orig_update
            out["x0_%d" % r] = obs.copy()
            out["costs_%d" % r] = cap["costs"]
            out["eps_sub_%d" % r] = cap["eps"][:, ::16, :].copy()     # (H, N/16, 1) post-clip
            out["act_%d" % r] = ctl.act_sequence.copy()
            out["u_%d" % r] = u.copy()
            out["newstate_%d" % r] = constate.copy()
            obs = model.pred(obs, u)
        save("mppi_" + tag, nx=nx, hidden=np.array(hidden), activation=act, mlp_seed=mseed,
             plain_norm=(ckind == "plain"), wsum=weight_checksum(p), N=N, H=H, sigma=sigma,
             lmda=lmda, bounds=np.array(bnd), Q=Q, R=R, F=F, goal=cost.get_goal(),
             np_seed=npseed, n_runs=n_runs, **out)


# -------------------------------------------------------------------------- iLQR
class RefCubic(Model):
    """The analytic test model wrapped in the reference's Model ABC."""

    def __init__(self, system):
        super().__init__(system)
        self.impl = CubicIntegrator(system)

    @property
    def state_dim(self):
        return self.impl.state_dim

    def traj_to_state(self, traj):
        return traj[-1].obs.copy()

    def update_state(self, state, new_ctrl, new_obs):
        return new_obs.copy()

    def pred(self, state, ctrl):
        return self.impl.pred(state, ctrl)

    def pred_batch(self, states, ctrls):
        return self.impl.pred_batch(states, ctrls)

    def pred_diff(self, state, ctrl):
        return self.impl.pred_diff(state, ctrl)

    def pred_diff_batch(self, states, ctrls):
        return self.impl.pred_diff_batch(states, ctrls)


ILQR_CASES = [
    # tag, model kind, nx, nu, hidden, act, mseed, H, bounded, costkind, x0scale
    ("cubic_free", "cubic", 2, 1, None, None, 0, 20, None, "plain", 0.5),
    ("cubic_bounded", "cubic", 2, 1, None, None, 0, 20, (-0.2, 0.2), "plain", 0.5),
    ("p64_tanh_free", "mlp", 2, 1, [64, 64], "tanh", 41, 20, None, "plain", 0.3),
    ("p64_tanh_bounded", "mlp", 2, 1, [64, 64], "tanh", 41, 20, (-0.5, 0.5), "dense", 0.3),
    ("hc6_relu_free", "mlp", 17, 6, [256, 256], "relu", 42, 50, None, "plain", 0.1),
    ("p64_tanh_clipped", "mlp", 2, 1, [64, 64], "tanh", 41, 20, (-0.1, 0.15), "plain", 0.3),
    ("hc6_relu_bounded", "mlp", 17, 6, [256, 256], "relu", 42, 50, (-0.25, 0.25), "plain", 0.1),
    ("hc6_tanh_free", "mlp", 17, 6, [256, 256], "tanh", 43, 50, None, "dense", 0.1),
]


def gen_ilqr():
    for tag, kind, nx, nu, hidden, act, mseed, H, bnd, ckind, x0s in ILQR_CASES:
        system = make_system(nx, nu, dt=0.05)
        if kind == "cubic":
            model, p = RefCubic(system), None
        else:
            model, p = ref_mlp(system, hidden, act, mseed, plain_norm=(ckind == "plain"))
        task = Task(system)
        cost = make_cost(system, ckind, mseed + 500)
        task.set_cost(cost)
        if bnd is not None:
            task.set_ctrl_bounds(np.full(nu, bnd[0]), np.full(nu, bnd[1]))
        ctl = IterativeLQR(system, task, model, H)
        x0 = np.random.default_rng(mseed + 77).uniform(-x0s, x0s, size=nx)
        calls = {"n": 0}
        orig = model.pred_diff_batch

        def counting(states, ctrls, orig=orig):
            calls["n"] += 1
            return orig(states, ctrls)
        model.pred_diff_batch = counting
        conv, states, ctrls, Ks, ks = quiet(ctl.compute_ilqr_default, x0, np.zeros((H, nu)),
                                            silent=True)
        n_refresh = calls["n"]
        u, newstate = quiet(ctl.run, np.concatenate([x0, np.zeros(nu)]), x0)
        model.pred_diff_batch = orig
        Q, R, F = cost.get_cost_matrices()
        extra = {} if p is None else {"wsum": weight_checksum(p)}
        save("ilqr_" + tag, kind=kind, nx=nx, nu=nu, hidden=np.array(hidden or []),
             activation=act or "", mlp_seed=mseed, plain_norm=(ckind == "plain"), H=H,
             bounded=bnd is not None, bounds=np.array(bnd if bnd else (0.0, 0.0)), dt=0.05,
             Q=Q, R=R, F=F, goal=cost.get_goal(), x0=x0, converged=conv, states=states,
             ctrls=ctrls, Ks=Ks, ks=ks, n_refresh=n_refresh, u=u, newstate=newstate, **extra)


# ------------------------------------------------------------------- closed loop
def gen_closed_loop():
    # MPPI (nu = 1) on a tanh surrogate, 20 steps; iLQR on the same surrogate.
    nx, hidden, act, mseed = 3, [48, 48], "tanh", 51
    system = make_system(nx, 1, dt=0.05)
    model, p = ref_mlp(system, hidden, act, mseed, plain_norm=True)
    task = Task(system)
    cost = make_cost(system, "dense", 600)
    task.set_cost(cost)
    task.set_ctrl_bound("u0", -1.0, 1.0)
    init = np.array([0.2, -0.1, 0.15])
    np.random.seed(4)
    ctl = quiet(MPPI, system, task, model, horizon=10, num_path=128, sigma=0.8, lmda=0.5)
    traj = quiet(simulate, ctl, init, sim_model=model, max_steps=20, silent=True)
    Q, R, F = cost.get_cost_matrices()
    common = dict(nx=nx, hidden=np.array(hidden), activation=act, mlp_seed=mseed,
                  wsum=weight_checksum(p), Q=Q, R=R, F=F, goal=cost.get_goal(), init=init)
    save("loop_mppi", np_seed=4, N=128, H=10, sigma=0.8, lmda=0.5, bounds=np.array([-1.0, 1.0]),
         obs=traj.obs, ctrls=traj.ctrls, score=cost(traj), **common)
    task2 = Task(system)
    task2.set_cost(cost)
    ctl2 = IterativeLQR(system, task2, model, 12)
    traj2 = quiet(simulate, ctl2, init, sim_model=model, max_steps=15, silent=True)
    save("loop_ilqr", H=12, dt=0.05, obs=traj2.obs, ctrls=traj2.ctrls, score=cost(traj2), **common)


# ------------------------------------------------- closed loop, eval_cfg's call shape
def gen_evalcfg():
    """The tuner's objective exactly as PipelineTuner.eval_cfg computes it
    (tuning/pipeline_tuner.py:213-258): ``task.set_num_steps(T)``; controller built, then
    ``controller.reset()``; ``simulate(controller, task.get_init_obs(), task.term_cond,
    sim_model=surrogate, max_steps=task.get_num_steps())``; ``task.get_cost()(traj)``; then the
    same again with a fresh controller against ``dynamics=truedyn``.  ``task.term_cond`` is
    ``len(traj) >= num_steps`` (tasks/task.py:41-53) and simulate() breaks on it after extending
    the trajectory (utils/simulation.py:52-64), so an episode has num_steps rows = num_steps - 1
    controls.  A third case sets a user termination condition that fires before num_steps.
    (The factories need ConfigSpace, which is absent: the controller is constructed directly with
    the hyper-parameters ``pipeline(cfg, task, trajs)`` would pass.)"""
    nx, hidden, act, mseed = 3, [48, 48], "tanh", 51
    system = make_system(nx, 1, dt=0.05)
    model, p = ref_mlp(system, hidden, act, mseed, plain_norm=True)
    cost = make_cost(system, "dense", 600)
    Q, R, F = cost.get_cost_matrices()
    init = np.array([0.25, -0.15, 0.1])
    common = dict(nx=nx, hidden=np.array(hidden), activation=act, mlp_seed=mseed,
                  wsum=weight_checksum(p), Q=Q, R=R, F=F, goal=cost.get_goal(), init=init)

    def truedyn(x, u):
        return model.pred(x, u)

    def eval_cfg(make_controller, task, with_truedyn=True):
        out = {}
        controller = make_controller()
        controller.reset()
        surr_traj = quiet(simulate, controller, task.get_init_obs(), task.term_cond, sim_model=model,
                          max_steps=task.get_num_steps(), silent=True)
        out["surr_obs"], out["surr_ctrls"] = surr_traj.obs, surr_traj.ctrls
        out["surr_cost"] = task.get_cost()(surr_traj)
        if with_truedyn:
            controller = make_controller()
            controller.reset()
            td_traj = quiet(simulate, controller, task.get_init_obs(), task.term_cond,
                            dynamics=truedyn, max_steps=task.get_num_steps(), silent=True)
            out["truedyn_obs"], out["truedyn_ctrls"] = td_traj.obs, td_traj.ctrls
            out["truedyn_cost"] = task.get_cost()(td_traj)
        return out

    def mppi_task(T):
        task = Task(system)
        task.set_cost(cost)
        task.set_ctrl_bound("u0", -1.0, 1.0)
        task.set_init_obs(init)
        task.set_num_steps(T)
        return task

    hyper = dict(horizon=9, num_path=96, sigma=0.7, lmda=0.6)
    meta = dict(N=hyper["num_path"], H=hyper["horizon"], sigma=hyper["sigma"], lmda=hyper["lmda"],
                bounds=np.array([-1.0, 1.0]))
    # -- MPPI, default termination ---------------------------------------------------------
    T = 14
    task = mppi_task(T)
    np.random.seed(6)
    out = eval_cfg(lambda: quiet(MPPI, system, task, model, **hyper), task)
    assert len(out["surr_obs"]) == T and len(out["truedyn_obs"]) == T
    save("loop_evalcfg_mppi", np_seed=6, num_steps=T, **meta, **out, **common)
    # -- MPPI, user termination condition that fires early ------------------------------------
    T, min_len, thresh = 40, 4, 0.12
    task = mppi_task(T)
    task.set_term_cond(lambda traj: len(traj) >= min_len and abs(traj[-1].obs[0]) < thresh)
    assert task.has_num_steps() and task.get_num_steps() == T
    np.random.seed(8)
    out = eval_cfg(lambda: quiet(MPPI, system, task, model, **hyper), task)
    n = len(out["surr_obs"])
    assert min_len < n < T - 5, n
    save("loop_evalcfg_term", np_seed=8, num_steps=T, term_min_len=min_len, term_thresh=thresh,
         **meta, **out, **common)
    # -- iLQR, default termination -----------------------------------------------------------
    T = 11
    task2 = Task(system)
    task2.set_cost(cost)
    task2.set_init_obs(init)
    task2.set_num_steps(T)
    out = eval_cfg(lambda: IterativeLQR(system, task2, model, 12), task2)
    assert len(out["surr_obs"]) == T
    save("loop_evalcfg_ilqr", num_steps=T, H=12, dt=0.05, **out, **common)


# ---------------------------------------------------------- sums of quadratic costs
def _term_arrays(cost):
    """Per-term Q, R, F, goal of a (sum of) QuadCost(s), in the sum's order."""
    terms = cost.costs if hasattr(cost, "costs") else [cost]
    mats = [t.get_cost_matrices() for t in terms]
    return dict(Qs=np.stack([m[0] for m in mats]), Rs=np.stack([m[1] for m in mats]),
                Fs=np.stack([m[2] for m in mats]), goals=np.stack([np.asarray(t.get_goal(), dtype=float) for t in terms]))


def sum_cost_of(system, kind, seed, task_goal=None):
    """Controller costs that are sums of quadratics, built by the reference's own classes:
      gauss    QuadCostFactory(cfg) + GaussRegFactory(cfg)  -- the factories themselves called with
               dict configurations, combined as SumCostFactory.__call__ does (sum_cost_factory.py:45-53:
               ``sum(costs, SumCost(system, []))``); goal of the second term = mean of the data
               (gauss_reg_factory.py:37-45), of the first = the task's goal
      dense    two dense, non-symmetric QuadCosts with different goals, both with a terminal part
      three    three terms, two of them sharing a goal
      samegoal two terms with the SAME goal: the reference's SumCost is then ``is_quad`` and its
               get_goal() returns a cost OBJECT (sum_cost.py:45-47)"""
    from autompc.costs import SumCost
    from autompc.costs.quad_cost_factory import QuadCostFactory
    from autompc.costs.gauss_reg_factory import GaussRegFactory
    no, nu = system.obs_dim, system.ctrl_dim
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        task = Task(system)
        goal = rng.normal(scale=0.2, size=no) if task_goal is None else task_goal
        task.set_cost(QuadCost(system, np.eye(no), np.eye(nu), np.eye(no), goal=goal))
        cfg = {}
        for n in system.observations:
            cfg[n + "_Q"] = float(10 ** rng.uniform(-1, 1.5))
            cfg[n + "_F"] = float(10 ** rng.uniform(-1, 1.5))
        for n in system.controls:
            cfg[n + "_R"] = float(10 ** rng.uniform(-3, -1))
        trajs = []
        for k in range(5):
            tr = ampc.zeros(system, 30)
            tr.obs[:] = rng.normal(scale=0.4, size=(30, no)) @ (np.eye(no) + 0.3 * rng.normal(size=(no, no))) \
                + rng.normal(scale=0.3, size=no)
            trajs.append(tr)
        c1 = QuadCostFactory(system)(cfg, task, trajs)
        c2 = GaussRegFactory(system)({"reg_weight": float(10 ** rng.uniform(-2, -0.5))}, task, trajs)
        return sum([c1, c2], SumCost(system, []))

    def dense(goal_scale):
        A, B = rng.normal(size=(no, no)), rng.normal(size=(no, no))
        Q = A @ A.T / no + 0.1 * rng.normal(size=(no, no))
        F = B @ B.T / no + 0.05 * rng.normal(size=(no, no))
        R = np.diag(rng.uniform(0.01, 0.1, size=nu)) + 0.004 * rng.normal(size=(nu, nu))
        return QuadCost(system, Q, R, F, goal=rng.normal(scale=goal_scale, size=no))
    if kind == "dense":
        return dense(0.2) + dense(0.5)
    if kind == "three":
        a, b = dense(0.3), dense(0.3)
        c = QuadCost(system, np.diag(rng.uniform(0.1, 1, size=no)), 0.02 * np.eye(nu), None, goal=a.get_goal())
        return a + b + c
    if kind == "samegoal":
        a = dense(0.3)
        b = QuadCost(system, np.diag(rng.uniform(0.1, 1, size=no)), 0.02 * np.eye(nu), np.eye(no), goal=a.get_goal())
        return a + b
    raise ValueError(kind)


def gen_sumcost():
    from autompc.costs import SumCost
    # -- the nine eval_* entry points of sums (sum_cost.py:49-82) ---------------------------------
    system = make_system(5, 3)
    rng = np.random.default_rng(2024)
    out = {}
    for kind in ("gauss", "dense", "three", "samegoal"):
        cost = sum_cost_of(system, kind, 700 + len(kind))
        assert isinstance(cost, SumCost)
        obs, ctrl = rng.normal(size=5), rng.normal(size=3)
        o, c, t = cost.eval_obs_cost_hess(obs), cost.eval_ctrl_cost_hess(ctrl), cost.eval_term_obs_cost_hess(obs)
        traj = ampc.zeros(system, 7)
        traj.obs[:] = rng.normal(size=(7, 5))
        traj.ctrls[:] = rng.normal(size=(7, 3))
        arrs = _term_arrays(cost)
        out.update({kind + "_" + k: v for k, v in arrs.items()})
        out.update({kind + "_obs": obs, kind + "_ctrl": ctrl, kind + "_is_quad": bool(cost.is_quad),
                    kind + "_obs_cost": cost.eval_obs_cost(obs), kind + "_ctrl_cost": cost.eval_ctrl_cost(ctrl),
                    kind + "_term_cost": cost.eval_term_obs_cost(obs),
                    kind + "_obs_c": o[0], kind + "_obs_j": o[1], kind + "_obs_h": o[2],
                    kind + "_ctrl_c": c[0], kind + "_ctrl_j": c[1], kind + "_ctrl_h": c[2],
                    kind + "_term_c": t[0], kind + "_term_j": t[1], kind + "_term_h": t[2],
                    kind + "_traj_obs": traj.obs, kind + "_traj_ctrls": traj.ctrls, kind + "_traj_cost": cost(traj)})
    save("cost_sum", **out)

    # -- MPPI (nu = 1) with such costs: mppi.py:73-82 evaluates them term by term ----------------
    cases = [
        # tag, nx, hidden, act, mlpseed, N, H, sigma, lmda, bounds, costkind, npseed
        ("gauss", 2, [64, 64], "relu", 61, 512, 20, 1.0, 1.0, (-2.0, 2.0), "gauss", 11),
        ("dense", 4, [32, 32], "tanh", 62, 300, 15, 0.6, 0.3, (-1.0, 1.5), "dense", 12),
        ("hc_gauss", 17, [256, 256], "relu", 63, 256, 20, 1.0, 1.0, (-1.0, 1.0), "gauss", 13),
        ("samegoal", 3, [48, 48], "tanh", 64, 200, 12, 0.8, 0.5, (-1.0, 1.0), "samegoal", 14),
    ]
    for tag, nx, hidden, act, mseed, N, H, sigma, lmda, bnd, ckind, npseed in cases:
        system = make_system(nx, 1)
        model, p = ref_mlp(system, hidden, act, mseed, plain_norm=True)
        cost = sum_cost_of(system, ckind, mseed + 300)
        task = Task(system)
        task.set_cost(cost)
        task.set_ctrl_bound("u0", bnd[0], bnd[1])
        np.random.seed(npseed)
        ctl = quiet(MPPI, system, task, model, horizon=H, num_path=N, sigma=sigma, lmda=lmda)
        out = {"act0": ctl.act_sequence.copy()}
        obs = np.random.default_rng(npseed + 99).uniform(-0.1, 0.1, size=nx)
        constate = np.concatenate([obs, np.zeros(1)])
        n_runs = 3
        for r in range(n_runs):
            cap = {}
            orig_update = ctl.update

            def spy(costs, eps, cap=cap, orig=orig_update):
                cap["costs"] = costs.copy()
                cap["eps"] = eps.copy()
                return orig(costs, eps)
            ctl.update = spy
            u, constate = ctl.run(constate, obs)
            ctl.update = orig_update
            out["x0_%d" % r] = obs.copy()
            out["costs_%d" % r] = cap["costs"]
            out["eps_sub_%d" % r] = cap["eps"][:, ::16, :].copy()
            out["act_%d" % r] = ctl.act_sequence.copy()
            out["u_%d" % r] = u.copy()
            out["newstate_%d" % r] = constate.copy()
            obs = model.pred(obs, u)
        save("mppi_sumcost_" + tag, nx=nx, hidden=np.array(hidden), activation=act, mlp_seed=mseed,
             plain_norm=True, wsum=weight_checksum(p), N=N, H=H, sigma=sigma, lmda=lmda,
             bounds=np.array(bnd), np_seed=npseed, n_runs=n_runs, is_quad=bool(cost.is_quad),
             **_term_arrays(cost), **out)

    # -- iLQR with such costs: ilqr.py:124-129 (objective), :159-174 (gradients / Hessians) -------
    icases = [
        # tag, nx, nu, hidden, act, mseed, H, bounds, costkind, x0scale
        ("p64_gauss", 2, 1, [64, 64], "tanh", 71, 20, None, "gauss", 0.3),
        ("p64_dense_bounded", 2, 1, [64, 64], "tanh", 71, 20, (-0.4, 0.4), "dense", 0.3),
        ("hc6_gauss", 17, 6, [256, 256], "relu", 72, 50, None, "gauss", 0.1),
        ("hc6_three_bounded", 17, 6, [256, 256], "tanh", 73, 50, (-0.25, 0.25), "three", 0.1),
        ("q3_samegoal", 3, 2, [48, 48], "tanh", 74, 15, None, "samegoal", 0.3),
    ]
    for tag, nx, nu, hidden, act, mseed, H, bnd, ckind, x0s in icases:
        system = make_system(nx, nu, dt=0.05)
        model, p = ref_mlp(system, hidden, act, mseed, plain_norm=True)
        cost = sum_cost_of(system, ckind, mseed + 400)
        task = Task(system)
        task.set_cost(cost)
        if bnd is not None:
            task.set_ctrl_bounds(np.full(nu, bnd[0]), np.full(nu, bnd[1]))
        ctl = IterativeLQR(system, task, model, H)
        x0 = np.random.default_rng(mseed + 77).uniform(-x0s, x0s, size=nx)
        calls = {"n": 0}
        orig = model.pred_diff_batch

        def counting(states, ctrls, orig=orig, calls=calls):
            calls["n"] += 1
            return orig(states, ctrls)
        model.pred_diff_batch = counting
        conv, states, ctrls, Ks, ks = quiet(ctl.compute_ilqr_default, x0, np.zeros((H, nu)), silent=True)
        n_refresh = calls["n"]
        u, newstate = quiet(ctl.run, np.concatenate([x0, np.zeros(nu)]), x0)
        model.pred_diff_batch = orig
        save("ilqr_sumcost_" + tag, n_refresh=n_refresh, kind="mlp", nx=nx, nu=nu, hidden=np.array(hidden), activation=act,
             mlp_seed=mseed, plain_norm=True, H=H, bounded=bnd is not None,
             bounds=np.array(bnd if bnd else (0.0, 0.0)), dt=0.05, x0=x0, converged=conv, states=states,
             ctrls=ctrls, Ks=Ks, ks=ks, u=u, newstate=newstate, wsum=weight_checksum(p),
             is_quad=bool(cost.is_quad), **_term_arrays(cost))

    # -- eval_cfg's call shape (pipeline_tuner.py:213-258) with a QuadCostFactory + GaussRegFactory
    #    controller cost; the TASK cost that scores the episode stays the task's own ---------------
    nx, hidden, act, mseed = 3, [48, 48], "tanh", 51
    system = make_system(nx, 1, dt=0.05)
    model, p = ref_mlp(system, hidden, act, mseed, plain_norm=True)
    task_cost = make_cost(system, "dense", 600)
    init = np.array([0.25, -0.15, 0.1])
    ctl_cost = sum_cost_of(system, "gauss", 910, task_goal=task_cost.get_goal())
    Q, R, F = task_cost.get_cost_matrices()

    def eval_cfg(make_controller, task):
        controller = make_controller()
        controller.reset()
        traj = quiet(simulate, controller, task.get_init_obs(), task.term_cond, sim_model=model,
                     max_steps=task.get_num_steps(), silent=True)
        return dict(surr_obs=traj.obs, surr_ctrls=traj.ctrls, surr_cost=task.get_cost()(traj),
                    ctl_cost_of_traj=ctl_cost(traj))

    def tasks(T, bounded):
        task = Task(system)                 # what the tuner scores with
        task.set_cost(task_cost)
        ctask = Task(system)                # what the pipeline hands the controller (pipeline.py:156-160)
        ctask.set_cost(ctl_cost)
        for t in (task, ctask):
            if bounded:
                t.set_ctrl_bound("u0", -1.0, 1.0)
            t.set_init_obs(init)
            t.set_num_steps(T)
        return task, ctask
    common = dict(nx=nx, hidden=np.array(hidden), activation=act, mlp_seed=mseed, wsum=weight_checksum(p),
                  Q=Q, R=R, F=F, goal=task_cost.get_goal(), init=init, **_term_arrays(ctl_cost))
    T = 13
    task, ctask = tasks(T, True)
    hyper = dict(horizon=9, num_path=96, sigma=0.7, lmda=0.6)
    np.random.seed(16)
    out = eval_cfg(lambda: quiet(MPPI, system, ctask, model, **hyper), task)
    assert len(out["surr_obs"]) == T
    save("loop_evalcfg_sumcost", np_seed=16, num_steps=T, N=hyper["num_path"], H=hyper["horizon"],
         sigma=hyper["sigma"], lmda=hyper["lmda"], bounds=np.array([-1.0, 1.0]), **out, **common)
    T = 10
    task, ctask = tasks(T, False)
    out = eval_cfg(lambda: IterativeLQR(system, ctask, model, 12), task)
    assert len(out["surr_obs"]) == T
    save("loop_evalcfg_sumcost_ilqr", num_steps=T, H=12, dt=0.05, **out, **common)


# ------------------------------------------------------------------- score terms
def gen_cost_terms():
    """Cost.__call__ of threshold / box / summed costs on a batch of trajectories."""
    from autompc.costs import ThresholdCost, BoxThresholdCost
    system = make_system(5, 3)
    rng = np.random.default_rng(77)
    goal = rng.normal(scale=0.3, size=5)
    A = rng.normal(size=(5, 5))
    Q, F = A @ A.T / 5 + 0.1 * rng.normal(size=(5, 5)), np.diag(rng.uniform(0.5, 2.0, size=5))
    R = np.diag(rng.uniform(0.01, 0.1, size=3)) + 0.002
    quad = QuadCost(system, Q, R, F, goal=goal)
    quad2 = QuadCost(system, np.eye(5), np.eye(3), 2 * np.eye(5), goal=-goal)
    thr_lo, thr_hi, thr = 1, 4, 0.9
    thresh = ThresholdCost(system, goal, [thr_lo, thr_hi], thr)
    limits = np.array([[-1.0, 1.2], [-np.inf, 0.8], [-0.7, np.inf], [-np.inf, np.inf], [-1.5, 1.5]])
    box = BoxThresholdCost(system, limits, goal=goal)
    B, T = 6, 9
    obs = rng.normal(scale=0.8, size=(B, T, 5))
    ctrls = rng.normal(size=(B, T, 3))
    obs[0, 3, 2] = goal[2] + thr          # exactly on the threshold: not counted (strict >)
    obs[1, 4, 0] = limits[0, 1]           # exactly on a box limit: inside
    costs = {"quad": quad, "thresh": thresh, "box": box, "sum_tb": thresh + box,
             "sum_all": quad + thresh + box + quad2}
    scores = {}
    for name, c in costs.items():
        out = np.zeros(B)
        for b in range(B):
            traj = ampc.zeros(system, T)
            traj.obs[:] = obs[b]
            traj.ctrls[:] = ctrls[b]
            out[b] = c(traj)
        scores["score_" + name] = out
    save("cost_terms", Q=Q, R=R, F=F, goal=goal, thr_range=np.array([thr_lo, thr_hi]), thr=thr,
         limits=limits, obs=obs, ctrls=ctrls, **scores)


# ----------------------------------------------------------------- linear models
def linear_train_trajs(system, n_traj=6, T=40, seed=123):
    """Synthetic training set: a damped, weakly nonlinear oscillator driven by random controls."""
    no, nu = system.obs_dim, system.ctrl_dim
    rng = np.random.default_rng(seed)
    S = rng.normal(size=(no, no))
    M = np.eye(no) + 0.1 * (-0.4 * np.eye(no) + 0.5 * (S - S.T))
    G = rng.normal(scale=0.3, size=(no, nu))
    trajs = []
    for _ in range(n_traj):
        traj = ampc.zeros(system, T)
        x = rng.uniform(-1.0, 1.0, size=no)
        for t in range(T):
            u = rng.uniform(-1.0, 1.0, size=nu)
            traj[t].obs[:] = x
            traj[t].ctrl[:] = u
            x = M @ x + 0.05 * np.sin(2.0 * x[::-1]) + G @ u
        trajs.append(traj)
    return trajs


def gen_linear():
    from autompc.sysid.arx import ARX
    from autompc.sysid.koopman import Koopman
    system = make_system(3, 1)
    trajs = linear_train_trajs(system)
    train_obs = np.stack([t.obs for t in trajs])
    train_ctrls = np.stack([t.ctrls for t in trajs])
    models = {
        "arx3": quiet(ARX, system, history=3),
        "koop_full": quiet(Koopman, system, method="lstsq", poly_basis="true", poly_degree=3,
                           trig_basis="true", trig_freq=2, product_terms="false"),
        "koop_lasso": quiet(Koopman, system, method="lasso", lasso_alpha=1e-4, poly_basis="true",
                            poly_degree=2, trig_basis="false", product_terms="false"),
    }
    rng = np.random.default_rng(9)
    cost = make_cost(system, "dense", 700)
    Q, R, F = cost.get_cost_matrices()
    for tag, model in models.items():
        quiet(model.train, trajs)
        ns = model.state_dim
        states = model.traj_to_states(trajs[1])[5:21].copy()
        ctrls = rng.uniform(-1, 1, size=(16, 1))
        d0 = model.pred_diff(states[0], ctrls[0])
        out = dict(A=model.A, B=model.B, state_dim=ns,
                   state_prefix7=model.traj_to_state(trajs[0][:7]),
                   state_prefix1=model.traj_to_state(trajs[0][:1]),
                   states_traj1=model.traj_to_states(trajs[1]),
                   upd_state=model.update_state(states[3], ctrls[3], trajs[2][9].obs),
                   upd_in_obs=trajs[2][9].obs, pb_states=states, pb_ctrls=ctrls,
                   pred_batch=model.pred_batch(states, ctrls), pred0=model.pred(states[0], ctrls[0]),
                   diff0_pred=d0[0], diff0_jx=d0[1], diff0_ju=d0[2])
        # closed loops through the reference's simulate(): MPPI and iLQR on the linear model
        task = Task(system)
        task.set_cost(cost)
        task.set_ctrl_bound("u0", -1.0, 1.0)
        init = np.array([0.4, -0.3, 0.2])
        np.random.seed(21)
        ctl = quiet(MPPI, system, task, model, horizon=8, num_path=64, sigma=0.6, lmda=0.7)
        out["mppi_act0"] = ctl.act_sequence.copy()
        tr = quiet(simulate, ctl, init, sim_model=model, max_steps=6, silent=True)
        out["mppi_obs"], out["mppi_ctrls"], out["mppi_score"] = tr.obs, tr.ctrls, cost(tr)
        task2 = Task(system)
        task2.set_cost(cost)
        ctl2 = IterativeLQR(system, task2, model, 10)
        one = ampc.zeros(system, 1)
        one[0].obs[:] = init
        x0 = model.traj_to_state(one)
        conv, st, ct, Ks, ks = quiet(ctl2.compute_ilqr_default, x0, np.zeros((10, 1)), silent=True)
        out.update(ilqr_x0=x0, ilqr_converged=conv, ilqr_states=st, ilqr_ctrls=ct, ilqr_Ks=Ks,
                   ilqr_ks=ks)
        if not tag.startswith("arx"):
            # (the reference's iLQR cannot be simulated on ARX: its traj_to_state omits the
            # control that run() strips again, ilqr.py:96-98 vs :278, so update_state is handed
            # a state one entry short)
            tr2 = quiet(simulate, ctl2, init, sim_model=model, max_steps=5, silent=True)
            out["ilqr_loop_obs"], out["ilqr_loop_ctrls"] = tr2.obs, tr2.ctrls
            out["ilqr_loop_score"] = cost(tr2)
        save("linear_" + tag, train_obs=train_obs, train_ctrls=train_ctrls, Q=Q, R=R, F=F,
             goal=cost.get_goal(), init=init, np_seed=21, N=64, H=8, sigma=0.6, lmda=0.7,
             ilqr_H=10, dt=system.dt, **out)


# ------------------------------------------------------------------------- SINDy
class _StandInLibrary:
    """What the reference hands to ``ps.CustomLibrary`` (sindy.py:146-150), kept as given."""

    def __init__(self, library_functions, function_names):
        self.functions, self.names = list(library_functions), list(function_names)


class _StandInSINDy:
    """Stand-in for ``pysindy.SINDy`` (pysindy~=1.0 is third-party and absent).  The ONLY pysindy
    behaviour restated here is CustomLibrary's feature enumeration -- for every library function,
    in list order, one feature per ``itertools.combinations(range(n_vars), n_args)`` -- and
    ``predict = Theta(x, u) @ coefficients.T`` with the variables ordered [states, controls] and
    named x0.., u0...  Feature VALUES come from the reference's own ``basis.func`` lambdas and
    feature NAMES from its own ``basis.name_func`` (basis_funcs.py:8-126).  ``fit`` installs the
    coefficients given in ``_StandInSINDy.next_coefficients`` instead of running STLSQ."""
    next_coefficients = None

    def __init__(self, feature_library, discrete_time, optimizer):
        self.lib, self.discrete_time = feature_library, discrete_time

    def fit(self, X, u=None, multiple_trajectories=False, t=None, x_dot=None):
        import itertools
        self.nx, self.nu = X[0].shape[1], u[0].shape[1]
        n = self.nx + self.nu
        self.combos = [(f, name, c) for f, name in zip(self.lib.functions, self.lib.names)
                       for c in itertools.combinations(range(n), f.__code__.co_argcount)]
        self.coef = np.array(_StandInSINDy.next_coefficients, dtype=np.float64)
        assert self.coef.shape == (self.nx, len(self.combos)), (self.coef.shape, len(self.combos))

    def get_feature_names(self):
        var = ["x%d" % i for i in range(self.nx)] + ["u%d" % i for i in range(self.nu)]
        return [name(*[var[j] for j in c]) for _, name, c in self.combos]

    def coefficients(self):
        return self.coef

    def predict(self, states, ctrls):
        V = np.concatenate([states, ctrls], axis=1)
        theta = np.stack([np.broadcast_to(f(*[V[:, j] for j in c]), (V.shape[0],))
                          for f, _, c in self.combos], axis=1)
        return theta @ self.coef.T


def ref_sindy(system, xi_fn, **hyper):
    """The reference's SINDy with basis_funcs built by its OWN train() (sindy.py:130-171) on top
    of the stand-in above; xi_fn(n_features) -> coefficients [nx, n_features]."""
    import itertools
    import autompc.sysid.sindy as ref_mod
    from autompc.sysid.sindy import SINDy
    ref_mod.ps.CustomLibrary = _StandInLibrary
    ref_mod.ps.SINDy = _StandInSINDy
    ref_mod.ps.STLSQ = lambda threshold: None
    model = SINDy(system, "lstsq", **hyper)
    nx, nu = system.obs_dim, system.ctrl_dim
    # number of features = what the reference's own basis list enumerates to
    probe = SINDy(system, "lstsq", **hyper)
    _StandInSINDy.next_coefficients = None

    class _Count(_StandInSINDy):
        def fit(self, X, u=None, **kw):
            n = X[0].shape[1] + u[0].shape[1]
            probe.n_feat = sum(len(list(itertools.combinations(range(n), f.__code__.co_argcount)))
                               for f in self.lib.functions)
    ref_mod.ps.SINDy = _Count
    dummy = ampc.zeros(system, 3)
    probe.train([dummy])
    ref_mod.ps.SINDy = _StandInSINDy
    _StandInSINDy.next_coefficients = xi_fn(probe.n_feat)
    model.train([dummy])
    return model


def sparse_xi(nx, n_feat, seed, identity=True, density=0.15, scale=0.05):
    rng = np.random.default_rng(seed)
    xi = (rng.random((nx, n_feat)) < density) * rng.normal(scale=scale, size=(nx, n_feat))
    if identity:
        xi[:, :nx] += np.eye(nx)
    return xi


SINDY_CASES = [
    # tag, nx, nu, hyper-parameters, identity part, seed
    ("c1_trig", 4, 1, dict(trig_basis="true", trig_freq=1, trig_interaction="true",
                           time_mode="discrete"), True, 61),
    ("poly3_trig2_cont", 3, 2, dict(poly_basis="true", poly_degree=3, trig_basis="true", trig_freq=2,
                                    trig_interaction=True, time_mode="continuous"), False, 62),
    ("poly4_disc", 2, 1, dict(poly_basis=True, poly_degree=4, time_mode="discrete"), True, 63),
    ("identity_cont", 5, 2, dict(time_mode="continuous"), False, 64),
    # polynomial cross terms (basis_funcs.py:27-93): degree 2 (x y) and degree 3 (x y^2, x^2 y, x y z)
    ("cross3", 3, 2, dict(poly_basis="true", poly_degree=3, poly_cross_terms="true",
                          time_mode="discrete"), True, 65),
    ("cross4_trig_cont", 2, 1, dict(poly_basis=True, poly_degree=4, poly_cross_terms=True, trig_basis=True,
                                    trig_freq=1, trig_interaction=True, time_mode="continuous"), False, 66),
]


def gen_sindy():
    """SINDy inference (sindy.py:173-244) incl. the name-lookup Jacobian; config 1 of BASELINE.json
    (CartPole-shaped SINDy model, MPPI 256 x 20, the reference's CPU path) and the reference's own
    SINDy + iLQR pairing (tests/test_pipeline.py:96-160)."""
    for tag, nx, nu, hyper, ident, seed in SINDY_CASES:
        system = make_system(nx, nu, dt=0.05)
        model = ref_sindy(system, lambda nf: sparse_xi(nx, nf, seed, identity=ident), **hyper)
        rng = np.random.default_rng(seed + 1000)
        states, ctrls = rng.normal(scale=0.7, size=(24, nx)), rng.normal(scale=0.7, size=(24, nu))
        pb = model.pred_batch(states, ctrls)
        db, jx, ju = model.pred_diff_batch(states, ctrls)
        d0 = model.pred_diff(states[0], ctrls[0])
        out = dict(nx=nx, nu=nu, dt=0.05, time_mode=hyper["time_mode"],
                   trig_freq=int(hyper.get("trig_freq", 0)) if hyper.get("trig_basis") else 0,
                   trig_interaction=bool(hyper.get("trig_interaction")) and bool(hyper.get("trig_basis")),
                   poly_degree=int(hyper.get("poly_degree", 1)) if hyper.get("poly_basis") else 1,
                   poly_cross_terms=bool(hyper.get("poly_cross_terms")) and bool(hyper.get("poly_basis")),
                   Xi=model.model.coefficients(), feature_names=np.array(model.model.get_feature_names()),
                   states=states, ctrls=ctrls, pred_batch=pb, diff_pred=db, diff_jx=jx, diff_ju=ju,
                   pred0=model.pred(states[0], ctrls[0]), diff0_pred=d0[0], diff0_jx=d0[1], diff0_ju=d0[2])
        if tag == "c1_trig":
            # BASELINE config 1: MPPI 256 samples x 20 horizon on the SINDy model, reference CPU path
            task = Task(system)
            Q, R, F = np.diag([1.0, 10.0, 0.1, 0.1]), 0.01 * np.eye(1), np.eye(4)
            cost = QuadCost(system, Q, R, F, goal=np.zeros(4))
            task.set_cost(cost)
            task.set_ctrl_bound("u0", -2.0, 2.0)
            np.random.seed(8)
            ctl = quiet(MPPI, system, task, model, horizon=20, num_path=256, sigma=1.0, lmda=1.0)
            out["mppi_act0"] = ctl.act_sequence.copy()
            obs = np.array([0.0, 0.2, 0.0, 0.0])
            constate = np.concatenate([obs, np.zeros(1)])
            for r in range(3):
                cap = {}
                orig_update = ctl.update

                def spy(costs, eps, cap=cap, orig=orig_update):
                    cap["costs"] = costs.copy()
                    return orig(costs, eps)
                ctl.update = spy
                u, constate = ctl.run(constate, obs)
                ctl.update = orig_update
                out["mppi_x0_%d" % r] = obs.copy()
                out["mppi_costs_%d" % r] = cap["costs"]
                out["mppi_act_%d" % r] = ctl.act_sequence.copy()
                out["mppi_u_%d" % r] = u.copy()
                obs = model.pred(obs, u)
            out.update(Q=Q, R=R, F=F, np_seed=8, N=256, H=20, sigma=1.0, lmda=1.0,
                       bounds=np.array([-2.0, 2.0]))
            # the reference's own pairing: iLQR on the SINDy model
            task2 = Task(system)
            task2.set_cost(cost)
            ctl2 = IterativeLQR(system, task2, model, 15)
            x0 = np.array([0.1, 0.3, -0.1, 0.05])
            conv, st, ct, Ks, ks = quiet(ctl2.compute_ilqr_default, x0, np.zeros((15, 1)), silent=True)
            out.update(ilqr_x0=x0, ilqr_H=15, ilqr_converged=conv, ilqr_states=st, ilqr_ctrls=ct,
                       ilqr_Ks=Ks, ilqr_ks=ks)
        save("sindy_" + tag, **out)


# ------------------------------------------------------- wide linear model (41 states)
def gen_linear_wide():
    """ARX with history 2 on a HalfCheetah-sized system (17 observations, 6 controls): model state
    2*17 + 6 + 1 = 41 entries -- the reference's arx.py:42-187 fit and prediction, and one
    compute_ilqr_default on it (ilqr.py:100-265).  (The reference's MPPI cannot run with 6
    controls, mppi.py:21-24.)"""
    from autompc.sysid.arx import ARX
    system = make_system(17, 6)
    trajs = linear_train_trajs(system, n_traj=8, T=60, seed=321)
    model = quiet(ARX, system, history=2)
    quiet(model.train, trajs)
    assert model.state_dim == 41
    rng = np.random.default_rng(19)
    states = model.traj_to_states(trajs[1])[5:29].copy()
    ctrls = rng.uniform(-1, 1, size=(24, 6))
    d0 = model.pred_diff(states[0], ctrls[0])
    cost = make_cost(system, "dense", 900)
    Q, R, F = cost.get_cost_matrices()
    task = Task(system)
    task.set_cost(cost)
    ctl = IterativeLQR(system, task, model, 12)
    one = ampc.zeros(system, 1)
    init = np.random.default_rng(4).uniform(-0.4, 0.4, size=17)
    one[0].obs[:] = init
    x0 = model.traj_to_state(one)
    conv, st, ct, Ks, ks = quiet(ctl.compute_ilqr_default, x0, np.zeros((12, 6)), silent=True)
    save("linear_arx2_wide", train_obs=np.stack([t.obs for t in trajs]),
         train_ctrls=np.stack([t.ctrls for t in trajs]), A=model.A, B=model.B, state_dim=41,
         pb_states=states, pb_ctrls=ctrls, pred_batch=model.pred_batch(states, ctrls),
         pred0=model.pred(states[0], ctrls[0]), diff0_pred=d0[0], diff0_jx=d0[1], diff0_ju=d0[2],
         Q=Q, R=R, F=F, goal=cost.get_goal(), init=init, ilqr_H=12, dt=system.dt, ilqr_x0=x0,
         ilqr_converged=conv, ilqr_states=st, ilqr_ctrls=ct, ilqr_Ks=Ks, ilqr_ks=ks)


# ------------------------------------------- wide linear models (65..256 states) + lifted closed loop
def gen_linear_wide2():
    """ARX with history 10 (the top of the reference's range, arx.py:27,37-45) on a HalfCheetah-sized
    system (18 observations, 6 controls): 10*18 + 9*6 + 1 = 235 model states -- fit, state construction,
    prediction, Jacobians.  The same history on 18 observations / ONE control (190 states) additionally
    runs the reference's MPPI through simulate() (the reference's MPPI needs ctrl_dim 1, mppi.py:21-24).
    Only the dense rows of [A | B] are stored (the other rows are the history shift, arx.py:121-148)."""
    from autompc.sysid.arx import ARX
    for tag, nu in (("arx10_hc", 6), ("arx10_nu1", 1)):
        system = make_system(18, nu)
        trajs = linear_train_trajs(system, n_traj=10, T=70, seed=77 + nu)
        model = quiet(ARX, system, history=10)
        quiet(model.train, trajs)
        ns = model.state_dim
        assert ns == 10 * 18 + 9 * nu + 1
        rng = np.random.default_rng(5 + nu)
        states = model.traj_to_states(trajs[1])[12:36].copy()
        ctrls = rng.uniform(-1, 1, size=(24, nu))
        d0 = model.pred_diff(states[0], ctrls[0])
        A, B = model.A, model.B
        probe = rng.integers(0, ns, size=(64, 2))
        out = dict(coeffs=np.concatenate([A[:18], B[:18]], axis=1), state_dim=ns,
                   A_probe_idx=probe, A_probe=A[probe[:, 0], probe[:, 1]], A_sum=A.sum(), A_abs_sum=np.abs(A).sum(),
                   B_abs_sum=np.abs(B).sum(), A_diff0_equal=bool(np.array_equal(d0[1], A)),
                   state_prefix12=model.traj_to_state(trajs[0][:12]), state_prefix1=model.traj_to_state(trajs[0][:1]),
                   pb_states=states, pb_ctrls=ctrls, pred_batch=model.pred_batch(states, ctrls),
                   pred0=model.pred(states[0], ctrls[0]), diff0_pred=d0[0])
        if nu == 1:
            cost = make_cost(system, "dense", 1234)
            Q, R, F = cost.get_cost_matrices()
            task = Task(system)
            task.set_cost(cost)
            task.set_ctrl_bound("u0", -1.0, 1.0)
            init = np.random.default_rng(3).uniform(-0.4, 0.4, size=18)
            np.random.seed(31)
            ctl = quiet(MPPI, system, task, model, horizon=8, num_path=96, sigma=0.6, lmda=0.7)
            out["mppi_act0"] = ctl.act_sequence.copy()
            tr = quiet(simulate, ctl, init, sim_model=model, max_steps=6, silent=True)
            out.update(mppi_obs=tr.obs, mppi_ctrls=tr.ctrls, mppi_score=cost(tr), Q=Q, R=R, F=F,
                       goal=cost.get_goal(), init=init, np_seed=31, N=96, H=8, sigma=0.6, lmda=0.7)
        save("linear_" + tag, train_obs=np.stack([t.obs for t in trajs]),
             train_ctrls=np.stack([t.ctrls for t in trajs]), **out)


def gen_evalcfg_koopman():
    """eval_cfg's call shape (pipeline_tuner.py:213-258) with the reference's MPPI on its Koopman model
    (poly + trig basis): the controller re-lifts every observation (koopman.py:166-168) while
    simulate() advances the lifted simulation state with the model's own prediction
    (simulation.py:52-58) -- the two differ, which is what the device loop has to reproduce.  A second
    episode simulates on an MLP surrogate (state = observation) with the same Koopman controller."""
    from autompc.sysid.koopman import Koopman
    system = make_system(3, 1)
    trajs = linear_train_trajs(system)
    model = quiet(Koopman, system, method="lstsq", poly_basis="true", poly_degree=3, trig_basis="true",
                  trig_freq=2, product_terms="false")
    quiet(model.train, trajs)
    sur, p = ref_mlp(system, [48, 48], "tanh", 51, plain_norm=True)
    cost = make_cost(system, "dense", 700)
    Q, R, F = cost.get_cost_matrices()
    init = np.array([0.35, -0.25, 0.2])
    T = 12
    task = Task(system)
    task.set_cost(cost)
    task.set_ctrl_bound("u0", -1.0, 1.0)
    task.set_init_obs(init)
    task.set_num_steps(T)
    hyper = dict(horizon=8, num_path=64, sigma=0.6, lmda=0.7)
    out = {}
    for tag, sim_model, seed in (("self", model, 41), ("mlp", sur, 42)):
        np.random.seed(seed)
        ctl = quiet(MPPI, system, task, model, **hyper)
        ctl.reset()
        tr = quiet(simulate, ctl, task.get_init_obs(), task.term_cond, sim_model=sim_model,
                   max_steps=task.get_num_steps(), silent=True)
        assert len(tr) == T
        out.update({tag + "_obs": tr.obs, tag + "_ctrls": tr.ctrls, tag + "_cost": cost(tr), tag + "_np_seed": seed})
    save("loop_evalcfg_koopman", A=model.A, B=model.B, state_dim=model.state_dim, Q=Q, R=R, F=F,
         goal=cost.get_goal(), init=init, num_steps=T, N=64, H=8, sigma=0.6, lmda=0.7,
         bounds=np.array([-1.0, 1.0]), mlp_seed=51, hidden=np.array([48, 48]), activation="tanh",
         wsum=weight_checksum(p), **out)


# ---------------------------------------------- MPPI on costs with threshold / box terms
def gen_mppi_indicator():
    """The reference's MPPI charges whatever Cost the task holds, term by term (mppi.py:73-82): a
    QuadCost + ThresholdCost sum, a bare BoxThresholdCost, and quad + threshold + box.  Thresholds /
    limits sit inside the cloud of rolled-out states, so some samples pay and some do not."""
    from autompc.costs import ThresholdCost, BoxThresholdCost
    cases = [
        # tag, nx, hidden, act, mlpseed, N, H, sigma, lmda, bounds, npseed
        ("quadthresh", 2, [64, 64], "tanh", 61, 256, 12, 1.0, 0.8, (-2.0, 2.0), 11),
        ("box", 2, [64, 64], "tanh", 62, 256, 12, 1.0, 1.0, (-2.0, 2.0), 12),
        ("all_hc", 17, [256, 256], "relu", 63, 128, 10, 1.0, 1.0, (-1.0, 1.0), 13),
    ]
    for tag, nx, hidden, act, mseed, N, H, sigma, lmda, bnd, npseed in cases:
        system = make_system(nx, 1)
        model, p = ref_mlp(system, hidden, act, mseed, plain_norm=True)
        rng = np.random.default_rng(mseed)
        obs0 = np.random.default_rng(npseed + 99).uniform(-0.1, 0.1, size=nx)
        goal = rng.normal(scale=0.05, size=nx)
        quad = QuadCost(system, np.diag(rng.uniform(0.5, 2.0, size=nx)), 0.05 * np.eye(1),
                        np.diag(rng.uniform(0.5, 2.0, size=nx)), goal=goal)
        # probe the spread of the rollouts once to place the thresholds (a separate controller: the
        # global stream is re-seeded below)
        np.random.seed(1000 + npseed)
        t0 = Task(system)
        t0.set_cost(quad)
        t0.set_ctrl_bound("u0", bnd[0], bnd[1])
        spread = {}
        orig = model.pred_batch

        def spy(states, ctrls, spread=spread, orig=orig):
            out = orig(states, ctrls)
            spread.setdefault("dev", []).append(np.abs(out - goal).max(axis=1))
            return out
        model.pred_batch = spy               # (MPPI binds model.pred_batch at construction, mppi.py:71)
        probe = quiet(MPPI, system, t0, model, horizon=H, num_path=N, sigma=sigma, lmda=lmda)
        probe.run(np.concatenate([obs0, np.zeros(1)]), obs0)
        del model.pred_batch                 # back to the class's method
        dev = np.concatenate(spread["dev"])
        thr = float(np.quantile(dev, 0.6))
        lo, hi = (0, nx) if nx == 2 else (2, 11)
        thresh = ThresholdCost(system, goal, [lo, hi], thr)
        limits = np.stack([goal - np.quantile(dev, 0.8), goal + np.quantile(dev, 0.7)], axis=1)
        limits[nx - 1, 0] = -np.inf
        if nx > 2:
            limits[3] = [-np.inf, np.inf]
        box = BoxThresholdCost(system, limits, goal=goal)
        cost = {"quadthresh": quad + thresh, "box": box, "all_hc": quad + thresh + box}[tag]
        task = Task(system)
        task.set_cost(cost)
        task.set_ctrl_bound("u0", bnd[0], bnd[1])
        np.random.seed(npseed)
        ctl = quiet(MPPI, system, task, model, horizon=H, num_path=N, sigma=sigma, lmda=lmda)
        out = {"act0": ctl.act_sequence.copy()}
        obs = obs0.copy()
        constate = np.concatenate([obs, np.zeros(1)])
        n_runs = 3
        for r in range(n_runs):
            cap = {}
            orig_update = ctl.update

            def spy_u(costs, eps, cap=cap, orig=orig_update):
                cap["costs"] = costs.copy()
                return orig(costs, eps)
            ctl.update = spy_u
            u, constate = ctl.run(constate, obs)
            ctl.update = orig_update
            out["x0_%d" % r] = obs.copy()
            out["costs_%d" % r] = cap["costs"]
            out["act_%d" % r] = ctl.act_sequence.copy()
            out["u_%d" % r] = u.copy()
            obs = model.pred(obs, u)
        # the indicator part really discriminates between samples
        ind = cap["costs"] - np.floor(cap["costs"])
        assert len(np.unique(np.round(cap["costs"] - ind))) > 2, tag
        Q, R, F = quad.get_cost_matrices()
        save("indmppi_" + tag, nx=nx, hidden=np.array(hidden), activation=act, mlp_seed=mseed, plain_norm=True,
             wsum=weight_checksum(p), N=N, H=H, sigma=sigma, lmda=lmda, bounds=np.array(bnd), np_seed=npseed,
             n_runs=n_runs, has_quad=tag != "box", has_thresh=tag != "box", has_box=tag != "quadthresh",
             Q=Q, R=R, F=F, goal=goal, thr_range=np.array([lo, hi]), thr=thr, limits=limits, **out)


# ------------------------------------------- eval_cfg with a controller model per configuration
def gen_evalcfg_twomodels():
    """eval_cfg builds the controller with pipeline(cfg, task, trajs), which instantiates (trains) a model
    PER CONFIGURATION when the pipeline has a model factory (pipeline.py:138-145, pipeline_tuner.py:213-215),
    and simulates it against the ONE surrogate.  Two configurations whose controller models differ in their
    weights (same architecture): model A is also the surrogate, model B is not.  MPPI (own numpy seed each)
    and iLQR (different horizons)."""
    nx, hidden, act = 3, [48, 48], "tanh"
    system = make_system(nx, 1, dt=0.05)
    model_a, pa = ref_mlp(system, hidden, act, 51, plain_norm=True)
    model_b, pb = ref_mlp(system, hidden, act, 52, plain_norm=True)
    cost = make_cost(system, "dense", 600)
    Q, R, F = cost.get_cost_matrices()
    init = np.array([0.25, -0.15, 0.1])

    def eval_cfg(make_controller, task):
        controller = make_controller()
        controller.reset()
        tr = quiet(simulate, controller, task.get_init_obs(), task.term_cond, sim_model=model_a,
                   max_steps=task.get_num_steps(), silent=True)
        return tr.obs, tr.ctrls, task.get_cost()(tr)

    out = {}
    T = 12
    task = Task(system)
    task.set_cost(cost)
    task.set_ctrl_bound("u0", -1.0, 1.0)
    task.set_init_obs(init)
    task.set_num_steps(T)
    hyper = dict(horizon=9, num_path=96, sigma=0.7, lmda=0.6)
    for tag, model, seed in (("a", model_a, 16), ("b", model_b, 17)):
        np.random.seed(seed)
        o, c, sc = eval_cfg(lambda: quiet(MPPI, system, task, model, **hyper), task)
        assert len(o) == T
        out.update({"mppi_%s_obs" % tag: o, "mppi_%s_ctrls" % tag: c, "mppi_%s_cost" % tag: sc,
                    "mppi_%s_np_seed" % tag: seed})
    T2 = 9
    task2 = Task(system)
    task2.set_cost(cost)
    task2.set_init_obs(init)
    task2.set_num_steps(T2)
    for tag, model, H in (("a", model_a, 12), ("b", model_b, 9)):
        o, c, sc = eval_cfg(lambda: IterativeLQR(system, task2, model, H), task2)
        assert len(o) == T2
        out.update({"ilqr_%s_obs" % tag: o, "ilqr_%s_ctrls" % tag: c, "ilqr_%s_cost" % tag: sc, "ilqr_%s_H" % tag: H})
    assert np.max(np.abs(out["mppi_a_obs"] - out["mppi_b_obs"])) > 1e-3
    assert np.max(np.abs(out["ilqr_a_obs"] - out["ilqr_b_obs"])) > 1e-3
    save("loop_evalcfg_twomodels", nx=nx, hidden=np.array(hidden), activation=act, mlp_seed_a=51, mlp_seed_b=52,
         wsum_a=weight_checksum(pa), wsum_b=weight_checksum(pb), Q=Q, R=R, F=F, goal=cost.get_goal(), init=init,
         num_steps_mppi=T, num_steps_ilqr=T2, dt=0.05, N=96, H=9, sigma=0.7, lmda=0.6, bounds=np.array([-1.0, 1.0]),
         **out)


# --------------------------------------------------- iLQR on a model with more than 64 states
def gen_ilqr_arx4():
    """The reference's ARX default history is 4 (arx.py:27,37-45,165-166): on the 18-observation /
    6-control HalfCheetah that is 4*18 + 3*6 + 1 = 91 model states.  compute_ilqr_default (ilqr.py:100-265)
    unbounded and with clipped controls, and IterativeLQR.run on the lifted state."""
    from autompc.sysid.arx import ARX
    system = make_system(18, 6)
    trajs = linear_train_trajs(system, n_traj=10, T=70, seed=404)
    model = quiet(ARX, system, history=4)
    quiet(model.train, trajs)
    ns = model.state_dim
    assert ns == 91
    cost = make_cost(system, "dense", 1404)
    Q, R, F = cost.get_cost_matrices()
    H = 14
    init = np.random.default_rng(14).uniform(-0.4, 0.4, size=18)
    x0 = model.traj_to_state(trajs[2][:9])                    # a state with a real history in it
    out = {}
    for tag, bnd in (("free", None), ("clip", (-0.15, 0.2))):
        task = Task(system)
        task.set_cost(cost)
        if bnd is not None:
            task.set_ctrl_bounds(np.full(6, bnd[0]), np.full(6, bnd[1]))
        ctl = IterativeLQR(system, task, model, H)
        conv, st, ct, Ks, ks = quiet(ctl.compute_ilqr_default, x0, np.zeros((H, 6)), silent=True)
        u, newstate = quiet(ctl.run, np.concatenate([x0, np.zeros(6)]), trajs[2][8].obs)
        out.update({tag + "_converged": conv, tag + "_states": st, tag + "_ctrls": ct, tag + "_Ks": Ks,
                    tag + "_ks": ks, tag + "_u": u, tag + "_newstate": newstate})
    assert np.max(np.abs(out["clip_ctrls"] - out["free_ctrls"])) > 1e-3
    save("wideilqr_arx4_hc", coeffs=np.concatenate([model.A[:18], model.B[:18]], axis=1), state_dim=ns, history=4,
         A_sum=model.A.sum(), A_abs_sum=np.abs(model.A).sum(), B_abs_sum=np.abs(model.B).sum(),
         Q=Q, R=R, F=F, goal=cost.get_goal(), H=H, dt=system.dt, x0=x0, run_obs=trajs[2][8].obs.copy(),
         clip_bounds=np.array([-0.15, 0.2]), **out)


# --------------------------------------------------------------------------- MLP fitting
MLPFIT_CASES = [
    # tag, nx, nu, hidden, activation, init seed, lr, epochs, batch, (n_traj, rows)
    ("p_tanh", 3, 2, [32, 24], "tanh", 7, 3e-3, 4, 64, (3, 51)),          # 150 rows: 2 full batches + 22
    ("hc_relu3", 17, 6, [48, 32, 16], "relu", 11, 1e-3, 3, 64, (4, 49)),   # 192 rows: no ragged batch
    ("p_selu1", 2, 1, [40], "selu", 100, 1e-2, 5, 32, (2, 41)),
]


def gen_mlpfit():
    """The reference's own MLP(...).train(trajs) on its torch CPU path (mlp.py:137-165, 177-217): the net as
    constructed (torch.manual_seed(seed) + nn.Linear defaults), the normalisers and the net after training
    (Adam, SmoothL1, DataLoader(shuffle=True)), plus its predictions on fresh points."""
    for tag, nx, nu, hidden, act, seed, lr, n_iter, n_batch, (n_traj, rows) in MLPFIT_CASES:
        system = make_system(nx, nu)
        rng = np.random.default_rng(seed + 3000)
        trajs = []
        for _ in range(n_traj):
            t = ampc.zeros(system, rows)
            t.obs[:] = 0.1 * rng.normal(size=(rows, nx)).cumsum(axis=0)
            t.ctrls[:] = rng.normal(size=(rows, nu))
            trajs.append(t)
        kw = {"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)}
        model = quiet(MLP, system, n_hidden_layers=len(hidden), nonlintype=act, n_train_iters=n_iter,
                      n_batch=n_batch, lr=lr, seed=seed, use_cuda=False, **kw)
        init = [p.detach().clone().numpy() for p in model.net.parameters()]     # w0, b0, w1, b1, ...
        quiet(model.train, trajs)
        final = [p.detach().clone().numpy() for p in model.net.parameters()]
        states, ctrls = rng.normal(size=(16, nx)), rng.normal(size=(16, nu))
        out = {"init_%d" % i: v for i, v in enumerate(init)}
        out.update({"final_%d" % i: v for i, v in enumerate(final)})
        save("mlpfit_" + tag, nx=nx, nu=nu, hidden=np.array(hidden), activation=act, seed=seed, lr=lr,
             n_train_iters=n_iter, n_batch=n_batch, obs=np.stack([t.obs for t in trajs]),
             ctrls=np.stack([t.ctrls for t in trajs]), xu_means=model.xu_means, xu_std=model.xu_std,
             dy_means=model.dy_means, dy_std=model.dy_std, states=states, ctrls_q=ctrls,
             pred=model.pred_batch(states, ctrls), torch_version=torch.__version__, **out)


GENERATORS = {"mlpfit": gen_mlpfit, "ilqr_arx4": gen_ilqr_arx4, "evalcfg_twomodels": gen_evalcfg_twomodels, "mppi_indicator": gen_mppi_indicator, "linear_wide": gen_linear_wide, "sindy": gen_sindy, "linear": gen_linear, "mlp": gen_mlp, "cost": gen_cost, "mppi": gen_mppi, "ilqr": gen_ilqr,
              "closed_loop": gen_closed_loop, "evalcfg": gen_evalcfg, "cost_terms": gen_cost_terms, "sumcost": gen_sumcost, "linear_wide2": gen_linear_wide2, "evalcfg_koopman": gen_evalcfg_koopman}

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(GENERATORS)):
        GENERATORS[name]()
