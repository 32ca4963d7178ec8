"""Rollout-kernel time of the c3 MPPI solve (HalfCheetah 2x256, 4096 x 30, f64) for every activation
the MLP model offers: how much the activation's epilogue costs next to the relu headline."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autompc_amd import _lib
from oracle import mlp as omlp

nx, nu, N, H = 17, 6, 4096, 30
for prec in ("f64", "f32"):
    for act in ("relu", "tanh", "sigmoid", "selu"):
        p = omlp.random_params(nx, nu, [256, 256], act, seed=1)
        h = _lib.Handle(0, prec)
        h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
        h.set_quad_costs(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx))
        h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
        plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
        rng = np.random.default_rng(0)
        plan.upload(rng.uniform(-0.1, 0.1, size=nx), np.zeros((H, nu)), rng.normal(size=(N, H, nu)))
        for _ in range(20):
            plan.solve()
        plan.set_timing(True)
        for _ in range(100):
            plan.solve()
        t = plan.timing()
        print("%s %-8s rollout %.4f ms  update %.4f ms" % (prec, act, t["rollout_ms"], t["update_ms"]))
        plan.close(); h.close()
