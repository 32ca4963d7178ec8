"""Per-candidate controller models (ampc_*_plan_set_models): a c5-style batch of 64 MPPI candidates x 49 control
steps on the HalfCheetah surrogate with ONE model against 8 distinct 2x256 models (same batch, same noise),
and the iLQR evaluator the same way.  Plans with a model table run the run-time-shape kernels
(csrc/mppi_kernels.hpp: EXT), so the single-model row is shown on both kernel families.
python tools/models_rate.py [n_candidates] [n_rows]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import MLP                                           # noqa: E402
from autompc_amd.synthetic import make_workload                       # noqa: E402
from autompc_amd.tuning import (CandidateEvaluator, IlqrCandidateEvaluator, random_candidates,   # noqa: E402
                                random_ilqr_candidates)

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200      # (eval_cfg episodes of the c5 workload: 200 rows)
system, task, model, spec = make_workload("c3", precision="f64", device=0)
task.set_num_steps(T)
rng = np.random.default_rng(0)
models = []
for k in range(8):                       # eight "trained" variants: the surrogate's weights, perturbed
    m = MLP(system, n_hidden_layers=2, hidden_size=256, nonlintype="relu")
    m.weights = [w * (1.0 + 0.05 * rng.normal(size=w.shape)) for w in model.weights]
    m.biases = [b.copy() for b in model.biases]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = model.xu_means, model.xu_std, model.dy_means, model.dy_std
    models.append(m)


def timed(ev, cands, **kw):
    ev.evaluate(cands[:8], **kw)
    t0 = time.perf_counter()
    s = ev.evaluate(cands, **kw)
    return time.perf_counter() - t0, s


cands = random_candidates(system, C, seed=0)
with8 = [dict(c, model=models[i % 8]) for i, c in enumerate(cands)]
ev = CandidateEvaluator(system, task, model)
for name, cs, env in (("MPPI, one model, shape-specialised kernels", cands, {}),
                      ("MPPI, one model, run-time-shape kernels", cands, {"AMPC_STATIC": "0", "AMPC_JIT": "0"}),
                      ("MPPI, 8 models in the batch", with8, {})):
    os.environ.update(env)
    dt, s = timed(ev, cs, seed=1)
    for k in env:
        del os.environ[k]
    print("%-48s %d candidates x %d control steps: %.3f s = %.0f solves/s (finite: %d)"
          % (name, C, T - 1, dt, C * (T - 1) / dt, int(np.isfinite(s).sum())))
ic = random_ilqr_candidates(system, C, seed=0)
for c in ic:
    c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25
i8 = [dict(c, model=models[i % 8]) for i, c in enumerate(ic)]
iev = IlqrCandidateEvaluator(system, task, model)
for name, cs, env in (("iLQR, one model, shape-specialised kernels", ic, {}),
                      ("iLQR, one model, run-time-shape kernels", ic, {"AMPC_STATIC": "0", "AMPC_JIT": "0"}),
                      ("iLQR, 8 models in the batch", i8, {})):
    os.environ.update(env)
    dt, s = timed(iev, cs)
    for k in env:
        del os.environ[k]
    print("%-48s %d candidates x %d control steps: %.3f s = %.0f solves/s (finite: %d)"
          % (name, C, T - 1, dt, C * (T - 1) / dt, int(np.isfinite(s).sum())))
