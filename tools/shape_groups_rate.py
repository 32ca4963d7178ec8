"""Evaluating a batch whose candidates carry controller models of DIFFERENT shapes (the full MLPFactory space:
1-4 hidden layers of 16-256 units, mlp.py:107-135): one plan per shape.  64 MPPI candidates x 50-row episodes on
the HalfCheetah surrogate with 1 / 16 / 64 distinct model shapes in the batch.  python tools/shape_groups_rate.py"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import MLP                                            # noqa: E402
from autompc_amd.synthetic import make_workload                        # noqa: E402
from autompc_amd.tuning import CandidateEvaluator, random_candidates   # noqa: E402
from autompc_amd.tuning.configs import sample_mlp_config               # noqa: E402

system, task, model, spec = make_workload("c3", precision="f64", device=0)
task.set_num_steps(50)
rng = np.random.default_rng(0)
cands = random_candidates(system, 64, seed=0)
for c in cands:
    c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25


def shaped(n_shapes, jit):
    ms = []
    for k in range(n_shapes):
        cfg = sample_mlp_config(rng)
        cfg.pop("lr")
        m = MLP(system, seed=k, **cfg)
        m.dy_std = np.full(system.obs_dim, 0.05)
        m.jit_kernels = jit          # False: what BatchPipelineTuner sets on the models it fits for one evaluation
        ms.append(m)
    return ms


ev = CandidateEvaluator(system, task, model)
ev.evaluate(cands[:4], seed=1)
for n_shapes, jit in ((1, True), (16, True), (64, True), (16, False), (64, False)):
    ms = shaped(n_shapes, jit)
    batch = [dict(c, model=ms[i % n_shapes]) for i, c in enumerate(cands)]
    t0 = time.perf_counter()
    s = ev.evaluate(batch, seed=1)
    t1 = time.perf_counter()
    s2 = ev.evaluate(batch, seed=1)
    t2 = time.perf_counter()
    assert np.array_equal(s, s2)
    print("%2d model shapes (%s) in a batch of 64 candidates x 49 control steps: first evaluation %.2f s, again %.2f s "
          "(finite %d)" % (n_shapes, "kernel builds started in the background" if jit else "one-evaluation models: no builds",
                           t1 - t0, t2 - t1, int(np.isfinite(s).sum())))
