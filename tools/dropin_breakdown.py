"""Where a hot MPPI.run() call spends its time on the HOST (c2 / c3, device and numpy-stream noise): the whole
Python call, the library call inside it (ctypes in -> out), and -- numpy mode -- the library call with the
pre-drawn next noise switched off.  python tools/dropin_breakdown.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autompc_amd import MPPI, _lib, zeros
from autompc_amd.synthetic import make_workload


def wrap(lib, name, acc):
    fn = getattr(lib, name)

    def timed(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        acc[0] += time.perf_counter() - t0
        acc[1] += 1
        return r
    return fn, timed


for name in ("c2", "c3"):
    system, task, model, spec = make_workload(name)
    for noise, env in (("device", {}), ("numpy", {}), ("numpy", {"AMPC_LEGACY_PREDRAW": "0"}),
                       ("numpy", {"AMPC_RUN_MAPPED": "0", "AMPC_LEGACY_PREDRAW": "0"})):
        os.environ.update(env)
        np.random.seed(0)
        ctl = MPPI(system, task, model, horizon=spec["horizon"], num_path=spec["num_path"], sigma=1.0, lmda=1.0,
                   noise=noise)
        obs = task.get_init_obs()
        one = zeros(system, 1)
        one.obs[0, :] = obs
        cs = ctl.traj_to_state(one)
        for _ in range(20):
            u, cs = ctl.run(cs, obs)
        lib = _lib.load()
        acc = [0.0, 0]
        sym = "ampc_mppi_run" if noise == "device" else "ampc_mppi_run_legacy"
        plan = ctl._device()
        orig, timed = wrap(plan.lib, sym, acc)
        setattr(plan.lib, sym, timed)
        n = 400
        t0 = time.perf_counter()
        for _ in range(n):
            u, cs = ctl.run(cs, obs)
        dt = time.perf_counter() - t0
        setattr(plan.lib, sym, orig)
        for k in env:
            del os.environ[k]
        inside = acc[0] / max(acc[1], 1)
        print("%-3s noise=%-7s %-52s call %.1f us = library %.1f us + Python around it %.1f us   (%.0f calls/s)"
              % (name, noise, " ".join("%s=%s" % kv for kv in env.items()) or "(defaults)", 1e6 * dt / n, 1e6 * inside,
                 1e6 * (dt / n - inside), n / dt))
