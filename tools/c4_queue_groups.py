"""Continuous batching in G independent queues (own handle / stream / host thread each): do the kernels of
one queue fill the CUs another queue's lock-step launches leave idle?  python tools/c4_queue_groups.py P B G [G ...]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

P, B = int(sys.argv[1]), int(sys.argv[2])
system, task, model, spec = make_workload("c3", precision="f64", device=0)
nx, nu = spec["nx"], spec["nu"]
Q, R, F = task.get_cost().get_cost_matrices()
x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=(P, nx))
for G in [int(a) for a in sys.argv[3:]]:
    hs, plans = [], []
    for g in range(G):
        h = _lib.Handle(0, "f64")
        model.stage_into(h)
        h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
        h.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
        hs.append(h)
        plans.append(_lib.IlqrPlan(h, B // G, 50, system.dt, clip_to_bounds=True))
    parts = np.array_split(np.arange(P), G)

    def run(g, out):
        out[g] = plans[g].solve_queue(x0[parts[g]], max_iter=50, gains=False, trajectories=False)
    for rep in range(2):
        out = [None] * G
        th = [threading.Thread(target=run, args=(g, out)) for g in range(G)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
    print("P %d  %d slots in %d queues: %.1f ms  %.0f solves/s" % (P, B, G, 1e3 * dt, P / dt))
    for pl in plans:
        pl.close()
    for h in hs:
        h.close()
