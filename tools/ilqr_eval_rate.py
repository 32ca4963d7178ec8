"""iLQR candidates on the HalfCheetah surrogate (the tuner's other controller, control/ilqr.py:31-41): the
device-resident episode chains (ampc_ilqr_closed_loop) against one batched solve + surrogate step per control
step.  python tools/ilqr_eval_rate.py [n_candidates] [n_steps]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd.synthetic import make_workload                      # noqa: E402
from autompc_amd.tuning import IlqrCandidateEvaluator, random_ilqr_candidates   # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
system, task, model, spec = make_workload("c3", precision="f64", device=0)
task.set_num_steps(T)
cands = random_ilqr_candidates(system, C, seed=0)
for c in cands:
    c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25     # gains 0.18 .. 10: O(1) costs
for name, kw in (("device-resident chains, one plan", {}),
                 ("device-resident chains, a plan per horizon", {"one_plan": False}),
                 ("host loop, one queue of solves per step", {"device_resident": False}),
                 ("host loop, a plan per horizon", {"device_resident": False, "one_plan": False})):
    ev = IlqrCandidateEvaluator(system, task, model, **kw)
    ev.evaluate(cands[:8])
    t0 = time.perf_counter()
    s = ev.evaluate(cands)
    dt = time.perf_counter() - t0
    extra = ""
    if hasattr(ev, "last_iterations") and kw.get("device_resident", True):
        extra = ", %.1f iLQR iterations per solve" % (ev.last_iterations.sum() / (C * (T - 1)))
    print("%-42s %d candidates x %d control steps: %.2f s = %.0f solves/s (finite scores: %d)%s"
          % (name, C, T - 1, dt, C * (T - 1) / dt, int(np.isfinite(s).sum()), extra))
