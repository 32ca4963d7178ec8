import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import mlp as omlp
from autompc_amd import _lib
nx, nu, H = 17, 6, 30
p = omlp.random_params(nx, nu, [256, 256], "relu", seed=5)
def make(par, B):
    os.environ["AMPC_LS4_PAR"] = par
    h = _lib.Handle(0, "f64")
    h.set_mlp(nx, nu, p["weights"], p["biases"], "relu", p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    h.set_quad_costs(np.eye(nx), 0.05 * np.eye(nu), 2 * np.eye(nx), np.zeros(nx))
    return h, _lib.IlqrPlan(h, B, H, 0.05)
bad = 0
for B in (1, 3, 7, 20, 60, 85):
    ha, pa = make("1", B); hb, pb = make("0", B)
    rng = np.random.default_rng(B)
    for rep in range(40 if B < 20 else 10):
        x0 = rng.uniform(-0.4, 0.4, size=(B, nx))
        a = pa.solve(x0, np.zeros((B, H, nu)), 25); b = pb.solve(x0, np.zeros((B, H, nu)), 25)
        for k in ("states", "ctrls", "iters", "converged", "objective"):
            if not np.array_equal(a[k], b[k]):
                bad += 1; print("MISMATCH B", B, "rep", rep, k)
    print("B", B, "ok", "par rows", pa.stats()["candidate_rows"], "seq rows", pb.stats()["candidate_rows"])
    pa.close(); pb.close(); ha.close(); hb.close()
print("mismatches:", bad)
