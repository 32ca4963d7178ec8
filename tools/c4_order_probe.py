"""Experiment: how much of the c4 drain (124 iterations launched for 82 of work) goes away if the queue admits the
problems that will run long FIRST, and whether anything known at admission predicts them.
    python tools/c4_order_probe.py [P] [B]        (on the GPU box)"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
system, task, model, spec = make_workload("c3", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
goal = task.get_cost().get_goal()
h = _lib.Handle(0, "f64")
model.stage_into(h)
h.set_quad_costs(Q, R, F, goal)
h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
rng = np.random.default_rng(0)
x0 = rng.uniform(-0.1, 0.1, size=(P, nx))
plan = _lib.IlqrPlan(h, B, 50, system.dt, clip_to_bounds=True)
q = plan.solve_queue(x0, max_iter=50, gains=False, trajectories=False)
it = q["iters"].astype(int)
print("iterations: mean %.1f, at the cap %.3f, launched %d" % (it.mean(), (it >= 50).mean(), plan.stats()["iterations"]))


def launches(order):
    """iterations launched when the problems are admitted in this order through B slots (a slot takes the next
    problem in the iteration after its problem finished)"""
    left = np.zeros(B, dtype=int)
    nxt, n = 0, 0
    while True:
        for s in range(B):
            if left[s] == 0 and nxt < len(order):
                left[s] = it[order[nxt]]
                nxt += 1
        if not left.any():
            return n
        left[left > 0] -= 1
        n += 1


# predictors known at admission
xs = x0.copy()
J0 = np.zeros(P)
for t in range(50):
    d = xs - goal
    J0 += np.einsum("pi,ij,pj->p", d, Q, d)
    xs = model.pred_batch(xs, np.zeros((P, nu)))
d = xs - goal
J0 += np.einsum("pi,ij,pj->p", d, F, d)
q1 = plan.solve_queue(x0, max_iter=2, gains=False, trajectories=False)
preds = {"|x0|": np.linalg.norm(x0, axis=1), "zero-control cost J0": J0, "objective after 2 iterations": q1["objective"],
         "relative drop in 2 iterations": (J0 - q1["objective"]) / J0}
from scipy.stats import spearmanr                              # noqa: E402
print("fifo: %d launches; ideal (sum / B): %.1f; longest first (oracle): %d; shortest first: %d"
      % (launches(np.arange(P)), it.sum() / B, launches(np.argsort(-it)), launches(np.argsort(it))))
for k, v in preds.items():
    r = spearmanr(v, it).correlation
    print("%-30s spearman %+.3f   admitted by it, descending: %d launches, ascending: %d"
          % (k, r, launches(np.argsort(-v)), launches(np.argsort(v))))
