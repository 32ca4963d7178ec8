"""Many-problem iLQR batches with the line-search kernel forced to four-row passes, forced to the twelve-row
tile, and picked per poll: every output must be identical bit for bit (ilqr_lsw.hpp).  Random MLP shapes of
the reference's configuration space, B = 96 .. 320 problems (the range where passes are NOT side by side),
batch solves and queues.  python tools/stress_lsw.py [n_rounds] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from oracle import mlp as omlp                                 # noqa: E402  (random parameters only)

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KEYS = ("states", "ctrls", "Ks", "ks", "objective", "iters", "converged", "status")
bad = 0
for rnd in range(n_rounds):
    nx = int(rng.choice([3, 8, 17, 24, 32]))
    nu = int(rng.choice([1, 2, 3, 6, 8]))
    nl = int(rng.integers(1, 4))
    hidden = [int(rng.choice([64, 100, 128, 192, 256])) for _ in range(nl)]
    if rnd % 3 == 0:
        nx, nu, hidden = 17, 6, [256, 256]                     # BASELINE config 4's shape (static kernels)
    act = str(rng.choice(["relu", "tanh", "sigmoid"]))
    B, H = int(rng.choice([96, 160, 256, 320])), int(rng.choice([10, 25, 50]))
    bounded = bool(rng.integers(0, 2))
    p = omlp.random_params(nx, nu, hidden, act, seed=int(rng.integers(1 << 30)))
    x0 = rng.uniform(-0.2, 0.2, size=(B + 40, nx))
    outs = {}
    for rb in ("1", "3", "0"):
        os.environ["AMPC_LS4_RB"] = rb
        h = _lib.Handle(0, "f64")
        h.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
        h.set_quad_costs(np.eye(nx), 0.05 * np.eye(nu), 2 * np.eye(nx), np.zeros(nx))
        if bounded:
            h.set_ctrl_bounds(-0.3 * np.ones(nu), 0.3 * np.ones(nu))
        plan = _lib.IlqrPlan(h, B, H, 0.05, clip_to_bounds=bounded)
        a = plan.solve(x0[:B], np.zeros((B, H, nu)), max_iter=20)
        rows = plan.stats()["candidate_rows"]
        q = plan.solve_queue(x0, max_iter=20)
        outs[rb] = (a, q, rows)
        plan.close(); h.close()
    same = all(np.array_equal(outs["1"][i][k], outs[rb][i][k]) for rb in ("3", "0") for i in (0, 1) for k in KEYS)
    bad += not same
    print("round %2d nx=%d nu=%d hidden=%s %s B=%d H=%d bounded=%d: iters %.1f, rows %d / %d / %d  %s"
          % (rnd, nx, nu, hidden, act, B, H, bounded, outs["1"][0]["iters"].mean(), outs["1"][2], outs["3"][2],
             outs["0"][2], "identical" if same else "DIFFERENT"))
print("%d rounds, %d with differences" % (n_rounds, bad))
sys.exit(1 if bad else 0)
