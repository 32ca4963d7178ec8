"""The fused softmin update of the four-row rollout (mppi_kernels.hpp: finish_update_if_last) publishes tile
partials with write-through stores and a relaxed ticket instead of agent-scope fences.  A stale read by the
finishing workgroup would change the result of a solve; so: ONE solve (same state, warm start, noise stream)
repeated n times -- the control and the updated sequence must never change -- and the same against the
combine-kernel path to rounding.  python tools/stress_fused_combine.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from autompc_amd import _lib                                   # noqa: E402
from autompc_amd.synthetic import make_workload                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
for name in ("c2", "arx"):
    system, task, model, spec = make_workload(name, precision="f64", device=0)
    res = {}
    for fused in ("1", "0"):
        os.environ["AMPC_FUSED_COMBINE"] = fused          # (off by default: measured slower, mppi_kernels.hpp)
        h = _lib.Handle(0, "f64")
        model.stage_into(h)
        Q, R, F = task.get_cost().get_cost_matrices()
        h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
        b = task.get_ctrl_bounds()
        h.set_ctrl_bounds(b[:, 0], b[:, 1])
        N, H = spec["num_path"], spec["horizon"]
        plan = _lib.MppiPlan(h, N, H, 1.0, 1.0)
        x0 = np.asarray(spec.get("x0", task.get_init_obs()), dtype=float)
        act = np.random.default_rng(0).normal(size=H * spec["nu"])
        ref = plan.run(x0, act, philox=(3, 7)).copy()
        a_ref = plan.download(act_seq=True, u=False)[0].copy()
        reps = n if fused == "1" else 2000
        t0, bad = time.perf_counter(), 0
        for k in range(reps):
            u = plan.run(x0, act, philox=(3, 7))
            if not np.array_equal(u, ref):
                bad += 1
            if k % 997 == 0 and not np.array_equal(plan.download(act_seq=True, u=False)[0], a_ref):
                bad += 1
        dt = time.perf_counter() - t0
        res[fused] = (ref, a_ref)
        print("%s fused=%s: %d repeated solves, %d differing results, %.1f us per call, kernel kind %d"
              % (name, fused, reps, bad, 1e6 * dt / reps, plan.kernel_kind()))
        plan.close()
        h.close()
    d = np.max(np.abs(res["1"][1] - res["0"][1])) / np.max(np.abs(res["0"][1]))
    print("%s fused vs combine kernel: relative difference of the updated sequence %.2e" % (name, d))
