"""Replay an MPPI case tools/fuzz_gpu.py saved on a violation (gpurun_out/fuzz_case_<n>.npz) under
different library settings.  Usage: python tools/fuzz_replay.py <case.npz>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = np.load(sys.argv[1], allow_pickle=True)
from autompc_amd import MLP, MPPI, QuadCost, System, Task
nx, nu, nl = int(d["nx"]), int(d["nu"]), int(d["n_layers"])
hidden = [int(v) for v in d["hidden"]]
system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)


def run(jit, mt, extra=None, details=True):
    os.environ["AMPC_JIT"], os.environ["AMPC_MT"] = jit, mt
    for k, v in (extra or {}).items():
        os.environ[k] = v
    m = MLP(system, n_hidden_layers=nl, nonlintype=str(d["act"]), precision=str(d["prec"]),
            **{"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)})
    m.weights = [d["W%d" % i].copy() for i in range(nl + 1)]
    m.biases = [d["b%d" % i].copy() for i in range(nl + 1)]
    m.xu_means, m.xu_std, m.dy_means, m.dy_std = d["xu_means"], d["xu_std"], d["dy_means"], d["dy_std"]
    task = Task(system)
    task.set_cost(QuadCost(system, d["Q"], d["R"], d["F"], goal=d["goal"]))
    task.set_ctrl_bounds(np.full(nu, float(d["lo"])), np.full(nu, float(d["hi"])))
    out = []
    for rep in range(4):
        np.random.seed(int(d["seed"]))
        ctl = MPPI(system, task, m, horizon=int(d["H"]), num_path=int(d["N"]), sigma=float(d["sigma"]), lmda=float(d["lmda"]))
        if jit == "1":
            ctl._device()
            ctl._handle.jit_wait()
        cs = np.concatenate([d["obs"], np.zeros(nu)])
        ctl.run(cs, d["obs"], return_details=True)
        c = ctl.last_costs
        err = np.abs(c - d["costs_ref"]) / np.abs(d["costs_ref"])
        out.append("%.1e[%s]" % (err.max(), ",".join(str(i) for i in np.nonzero(err > 1e-9)[0][:6])))
        kind, rows = ctl._device().kernel_kind(), ctl._device().info()["samples_per_wg"]
    for k in (extra or {}):
        os.environ.pop(k, None)
    print("jit %s MT %s %s kind %d rows %d: cost error per repetition %s" % (jit, mt, extra or "", kind, rows, " ".join(out)), flush=True)


for jit in ("1", "0"):
    for mt in ("1", "0", "2"):
        run(jit, mt)
run("1", "1", {"AMPC_PINGPONG": "0"})
run("1", "1", {"AMPC_FUSED_UPDATE": "0"})
run("1", "1", {"AMPC_DENSE_COST": "1"})
