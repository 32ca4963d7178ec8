"""Synthetic workloads of the BASELINE.json shapes (no datasets / checkpoints exist offline).

Weights follow torch.nn.Linear's default initialisation distribution
(U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases), drawn from a seeded numpy
generator; normalisers are mean 0 / std 1 on the input and mean 0 / std 0.1 on the output, which
keeps 30-step rollouts bounded.
"""
import numpy as np

from .costs import QuadCost
from .sysid import ARX, MLP, SINDy
from .system import System
from .task import Task

WORKLOADS = {
    # name: nx, nu, hidden, num_path, horizon, ctrl bound
    "c2": dict(label="Pendulum, MLP 2x64, MPPI 1024 samples x 30 horizon",
               nx=2, nu=1, hidden=[64, 64], num_path=1024, horizon=30, bound=2.0),
    "c3": dict(label="HalfCheetah (17-dim state, 6-dim ctrl), MLP 2x256, MPPI 4096 samples x 30 horizon",
               nx=17, nu=6, hidden=[256, 256], num_path=4096, horizon=30, bound=1.0),
    # BASELINE config 1 (the reference's own CPU-runnable case): CartPole-sized SINDy library model
    # (identity + trig(freq 1) + trig interaction terms, random sparse coefficients), MPPI 256 x 20.
    "c1": dict(label="CartPole (4-dim state, 1 ctrl), SINDy library model, MPPI 256 samples x 20 horizon",
               nx=4, nu=1, num_path=256, horizon=20, bound=20.0),
    # a linear model through the same rollout kernel.  CartPole-sized ARX,
    # history 4 => model state 4*4 + 3*1 + 1 = 20 entries, cost on the 4 observations.
    "arx": dict(label="CartPole-sized ARX (history 4, 20-dim model state, 4 obs, 1 ctrl), "
                      "MPPI 1024 samples x 30 horizon",
                obs=4, nu=1, history=4, num_path=1024, horizon=30, bound=1.0),
}


def random_mlp_params(nx, nu, hidden, seed=0, dy_std=0.1):
    rng = np.random.default_rng(seed)
    dims = [nx + nu] + list(hidden) + [nx]
    weights, biases = [], []
    for fan_in, fan_out in zip(dims[:-1], dims[1:]):
        bound = 1.0 / np.sqrt(fan_in)
        weights.append(rng.uniform(-bound, bound, size=(fan_out, fan_in)))
        biases.append(rng.uniform(-bound, bound, size=(fan_out,)))
    return dict(weights=weights, biases=biases, xu_means=np.zeros(nx + nu),
                xu_std=np.ones(nx + nu), dy_means=np.zeros(nx), dy_std=np.full(nx, dy_std))


def _make_arx_workload(spec, precision, device, seed):
    no, nu, k = spec["obs"], spec["nu"], spec["history"]
    system = System(["x%d" % i for i in range(no)], ["u%d" % i for i in range(nu)], dt=0.05)
    model = ARX(system, history=k, precision=precision, device=device)
    rng = np.random.default_rng(seed)
    # a stable random regression: obs' = 0.9 obs + small lag / control / constant terms
    coeffs = rng.normal(scale=0.03, size=(no, model._get_fvec_size()))
    coeffs[:, :no] += 0.9 * np.eye(no)
    coeffs[:, -nu:] = rng.normal(scale=0.2, size=(no, nu))
    model.set_parameters({"coeffs": coeffs})
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(no), 0.01 * np.eye(nu), np.eye(no), goal=np.zeros(no)))
    task.set_ctrl_bounds(np.full(nu, -spec["bound"]), np.full(nu, spec["bound"]))
    init = rng.uniform(-0.1, 0.1, size=no)
    task.set_init_obs(init)
    from .trajectory import zeros
    one = zeros(system, 1)
    one.obs[0, :] = init
    return system, task, model, dict(spec, nx=model.state_dim, hidden=[], linear=(model.A, model.B),
                                     x0=model.traj_to_state(one))


def _make_sindy_workload(spec, precision, device, seed):
    nx, nu = spec["nx"], spec["nu"]
    system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)
    model = SINDy(system, trig_basis=True, trig_freq=1, trig_interaction=True, time_mode="discrete",
                  precision=precision, device=device)
    rng = np.random.default_rng(seed)
    nf = model.coefficients.shape[1]
    Xi = np.zeros((nx, nf))
    Xi[:, :nx] = np.eye(nx)                       # x' = x + (weak coupling, sparse nonlinear terms)
    Xi[0, 2] = Xi[1, 3] = 0.05
    Xi += (rng.random(Xi.shape) < 0.1) * rng.normal(scale=0.02, size=Xi.shape)
    Xi[2, nx], Xi[3, nx] = 0.1, -0.08             # the control enters the velocities
    model.set_coefficients(Xi)
    task = Task(system)
    task.set_cost(QuadCost(system, np.diag([1.0, 10.0, 0.1, 0.1]), 0.01 * np.eye(nu), np.eye(nx)))
    task.set_ctrl_bounds(np.full(nu, -spec["bound"]), np.full(nu, spec["bound"]))
    task.set_init_obs(np.array([0.0, 0.2, 0.0, 0.0]))
    return system, task, model, dict(spec, hidden=[], sindy=dict(Xi=Xi, trig_freq=1, trig_interaction=True,
                                                                  poly_degree=1, n_feat=nf))


def make_workload(name, precision="f64", device=0, seed=0):
    """(system, task, model, spec) for a named BASELINE configuration."""
    spec = WORKLOADS[name]
    if name == "arx":
        return _make_arx_workload(spec, precision, device, seed)
    if name == "c1":
        return _make_sindy_workload(spec, precision, device, seed)
    nx, nu = spec["nx"], spec["nu"]
    system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)
    p = random_mlp_params(nx, nu, spec["hidden"], seed=seed)
    kw = {"hidden_size_%d" % (i + 1): h for i, h in enumerate(spec["hidden"])}
    model = MLP(system, n_hidden_layers=len(spec["hidden"]), nonlintype="relu",
                precision=precision, device=device, **kw)
    model.weights, model.biases = p["weights"], p["biases"]
    model.xu_means, model.xu_std = p["xu_means"], p["xu_std"]
    model.dy_means, model.dy_std = p["dy_means"], p["dy_std"]
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), goal=np.zeros(nx)))
    task.set_ctrl_bounds(np.full(nu, -spec["bound"]), np.full(nu, spec["bound"]))
    task.set_init_obs(np.random.default_rng(seed).uniform(-0.1, 0.1, size=nx))
    return system, task, model, dict(spec, params=p)
