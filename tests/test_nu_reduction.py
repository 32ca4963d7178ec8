"""SURVEY.md 8(c) / BASELINE.md 3: the property that pins the nu > 1 generalisation.  The reference
raises for ctrl_dim > 1 (its noise is hard-coded to one control dimension, mppi.py:21-24), so a
nu = 3 problem whose three control dimensions carry IDENTICAL noise, weights, bounds and cost must
reduce to the reference's nu = 1 recurrence (mppi.py:120-152) and reproduce the nu = 1 golden:
  model   the u column of W1 (and its normaliser) replicated three times at one third weight
  cost    R -> (R / 3) I_3          sum_j (R/3) u^2 = R u^2
  sigma   -> 3 sigma                (lmda/sigma) sum_j A_j eps_j = (lmda/sigma_1) A eps
  noise   the golden's legacy-stream draws replicated over the three dimensions
Oracle here (CPU); the same construction runs on the device in tests/test_gpu_mppi.py."""
import numpy as np

from conftest import golden
from helpers import check_weights, golden_params, make_system, rel_err
from oracle.costs import QuadCostOracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle


def widen_to_three_controls(p, nx):
    """nu = 1 MLP parameters -> the equivalent nu = 3 model for identical control inputs."""
    q = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in p.items()}
    W0 = p["weights"][0]
    q["weights"] = [np.concatenate([W0[:, :nx]] + [W0[:, nx:nx + 1] / 3.0] * 3, axis=1)] + \
                   [w.copy() for w in p["weights"][1:]]
    q["biases"] = [b.copy() for b in p["biases"]]
    q["xu_means"] = np.concatenate([p["xu_means"][:nx]] + [p["xu_means"][nx:nx + 1]] * 3)
    q["xu_std"] = np.concatenate([p["xu_std"][:nx]] + [p["xu_std"][nx:nx + 1]] * 3)
    return q


def reduction_problem(name="mppi_hc_nu1"):
    g = golden(name)
    nx, N, H = int(g["nx"]), int(g["N"]), int(g["H"])
    p1 = golden_params(nx, 1, g["hidden"], g["activation"], g["mlp_seed"], bool(g["plain_norm"]))
    check_weights(p1, g)
    p3 = widen_to_three_controls(p1, nx)
    scale = np.sqrt(float(g["sigma"]))
    np.random.seed(int(g["np_seed"]))
    act0 = np.random.normal(scale=scale, size=(H, 1))           # MPPI.__init__ draw (mppi.py:97-99)
    eps = [np.random.normal(scale=scale, size=(N, H, 1)) for _ in range(3)]   # one per run()
    return g, nx, N, H, p3, np.repeat(act0, 3, axis=1), [np.repeat(e, 3, axis=2) for e in eps]


def test_three_identical_controls_reduce_to_the_nu1_golden():
    g, nx, N, H, p3, act0, eps = reduction_problem()
    system = make_system(nx, 3)
    model = MLPOracle(system, p3)
    cost = QuadCostOracle(g["Q"], g["R"][0, 0] / 3.0 * np.eye(3), g["F"], g["goal"])
    bounds = np.tile(g["bounds"], (3, 1))
    ctl = MPPIOracle(model, cost, bounds, horizon=H, num_path=N, sigma=3.0 * float(g["sigma"]),
                     lmda=float(g["lmda"]))
    ctl.act_sequence = act0.copy()
    obs = g["x0_0"].copy()
    cs = np.concatenate([obs, np.zeros(3)])
    for r in range(3):
        np.testing.assert_allclose(obs, g["x0_%d" % r], rtol=1e-9, atol=1e-12)
        u, cs = ctl.run(cs, obs, eps_nhu=eps[r])
        assert rel_err(ctl.last_costs, g["costs_%d" % r]) < 1e-9
        for j in range(3):      # every control dimension follows the nu = 1 recurrence
            assert rel_err(ctl.act_sequence[:, j:j + 1], g["act_%d" % r]) < 1e-9
            assert rel_err(u[j:j + 1], g["u_%d" % r]) < 1e-9
        obs = model.pred(obs, u)
