import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import mlp as omlp
def run(ls4, hidden, act, nx=16, nu=6, B=1):
    os.environ["AMPC_LS4"] = ls4
    from autompc_amd import _lib
    p = omlp.random_params(nx, nu, hidden, act, seed=3)
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.05 * np.eye(nu), 2 * np.eye(nx), np.zeros(nx))
    plan = _lib.IlqrPlan(h, B, 50, 0.05)
    x0 = np.random.default_rng(0).uniform(-0.3, 0.3, size=(B, nx))
    plan.solve(x0, np.zeros((B, 50, nu)), 10)
    plan.set_timing(True)
    out = plan.solve(x0, np.zeros((B, 50, nu)), 30)
    t = plan.timing()
    plan.close(); h.close()
    return t["iter_ms"], int(out["iters"][0])
for hidden in ([256, 256], [128, 128], [192, 192], [256, 256, 256], [64, 64]):
    for act in ("relu", "tanh"):
        a = run("1", hidden, act); b = run("0", hidden, act)
        print(hidden, act, "ls4 %.3f ms  general %.3f ms  (iters %d/%d)" % (a[0], b[0], a[1], b[1]))
