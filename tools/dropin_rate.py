"""Rate of the drop-in Controller.run() call itself (host buffers in, host control out; what a
`simulate` loop sees), for both noise modes.  Not the headline metric: bench.py times the
device-resident solve."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autompc_amd import MPPI, IterativeLQR
from autompc_amd.synthetic import make_workload

for name in ("c3", "c2", "arx"):
    system, task, model, spec = make_workload(name)
    only = os.environ.get("DROPIN_ONLY")
    for noise in ("device", "numpy", "numpy_host"):        # "numpy" is the default mode of MPPI()
        if only and (name, noise) != tuple(only.split(":")):
            continue
        np.random.seed(0)
        ctl = MPPI(system, task, model, horizon=spec["horizon"], num_path=spec["num_path"], sigma=1.0,
                   lmda=1.0, noise=noise)
        obs = task.get_init_obs()
        from autompc_amd import zeros
        one = zeros(system, 1)
        one.obs[0, :] = obs
        cs = ctl.traj_to_state(one)
        n = 20 if noise == "numpy_host" else 300
        for _ in range(5):
            u, cs = ctl.run(cs, obs)
        if ctl._handle is not None and ctl._handle.jit_status()[0] == 1:
            ctl._handle.jit_wait()          # an unregistered shape (arx): the rate a controller reaches once its
            for _ in range(5):              # run-time compiled kernels are ready (it switches over by itself)
                u, cs = ctl.run(cs, obs)
        t0 = time.perf_counter()
        for _ in range(n):
            u, cs = ctl.run(cs, obs)
        dt = time.perf_counter() - t0
        print("%-4s MPPI.run noise=%-12s %8.1f calls/s  (%.3f ms per call)" % (name, noise, n / dt, 1e3 * dt / n))
if os.environ.get("DROPIN_ONLY"):
    sys.exit(0)
system, task, model, spec = make_workload("c3")
from autompc_amd import QuadCost, Task
t2 = Task(system)
t2.set_cost(task.get_cost())
ctl = IterativeLQR(system, t2, model, 50)
obs = task.get_init_obs()
cs = np.concatenate([obs, np.zeros(system.ctrl_dim)])
u, cs = ctl.run(cs, obs)
t0 = time.perf_counter()
for _ in range(5):
    u, cs = ctl.run(cs, obs)
dt = time.perf_counter() - t0
print("c4   IterativeLQR.run (one problem, %d iterations)  %.1f calls/s  (%.1f ms per call)"
      % (ctl.last_iters, 5 / dt, 1e3 * dt / 5))
