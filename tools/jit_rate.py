"""Kernels compiled at run time for an unregistered model shape against the run-time-shape kernels
and against a registered shape of the same size: four-row line search time per
iteration and MPPI rollout time, before / after the shape plugin is ready.
    python tools/jit_rate.py          (on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autompc_amd import _lib                       # noqa: E402
from oracle import mlp as omlp                     # noqa: E402


def handle(nx, nu, hidden, act):
    p = omlp.random_params(nx, nu, hidden, act, seed=3)
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.05 * np.eye(nu), 2 * np.eye(nx), np.zeros(nx))
    h.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
    return h


def ilqr_ms(h, nx, nu, B=1):
    plan = _lib.IlqrPlan(h, B, 50, 0.05)
    x0 = np.random.default_rng(0).uniform(-0.3, 0.3, size=(B, nx))
    plan.solve(x0, np.zeros((B, 50, nu)), 10)
    plan.set_timing(True)
    plan.solve(x0, np.zeros((B, 50, nu)), 30)
    t = plan.timing()
    kind = plan.kernel_kind()
    plan.close()
    return kind, t


def mppi_ms(h, nx, nu, N=4096, H=30):
    plan = _lib.MppiPlan(h, [N], [H], [1.0], [1.0])
    plan.upload(np.zeros((1, nx)), np.zeros(H * nu), None)
    plan.set_outputs(False)
    for i in range(20):
        plan.generate_eps(0, i)
        plan.solve()
    plan.set_timing(True)
    for i in range(100):
        plan.generate_eps(0, 100 + i)
        plan.solve()
    t = plan.timing()
    kind = plan.kernel_kind()
    plan.close()
    return kind, t["rollout_ms"]


KIND = {0: "run-time shape", 1: "registered", 2: "run-time compiled"}
for label, nx, nu, hidden, act in [("registered 17/6 2x256", 17, 6, [256, 256], "relu"),
                                   ("unregistered 16/6 2x256", 16, 6, [256, 256], "relu"),
                                   ("unregistered 16/6 3x192", 16, 6, [192, 192, 192], "relu"),
                                   ("unregistered 9/4 4x128 tanh", 9, 4, [128, 128, 128, 128], "tanh")]:
    os.environ["AMPC_JIT"] = "0"
    h = handle(nx, nu, hidden, act)
    k0, t0 = ilqr_ms(h, nx, nu)
    m0 = mppi_ms(h, nx, nu)
    h.close()
    os.environ["AMPC_JIT"] = "1"
    h = handle(nx, nu, hidden, act)
    t_wait = time.perf_counter()
    h.jit_wait()
    t_wait = time.perf_counter() - t_wait
    k1, t1 = ilqr_ms(h, nx, nu)
    m1 = mppi_ms(h, nx, nu)
    h.close()
    print("%-30s  line search %.3f ms (%s) -> %.3f ms (%s)   sweep %.3f -> %.3f ms   rollout 4096x30 %.3f -> %.3f ms"
          "   [plugin ready after %.1f s]" % (label, t0["iter_ms"], KIND[k0], t1["iter_ms"], KIND[k1],
                                              t0["riccati_ms"], t1["riccati_ms"], m0[1], m1[1], t_wait))
