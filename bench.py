#!/usr/bin/env python3
"""Headline benchmark: MPC solves/sec of the MPPI / iLQR inner solve on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c1|arx|c4|c5] [--precision f64|f32]

One "step" = one pass of the hot path over one batch of synthetic input:
  c3 (default, headline)  one complete MPPI solve of BASELINE.json config 3 (HalfCheetah-shaped MLP
      2x256, 4096 samples x 30 horizon): fresh device-resident noise, batched surrogate rollout with
      stage / action / terminal costs, softmin weighting and warm-start update -- everything
      MPPI.run() does between receiving an observation and returning a control
      (reference: autompc/control/mppi.py:120-168).  c2 / c1 / arx: the same for the other models.
  c4  --batch independent iLQR solves (HalfCheetah MLP, horizon 50; reference ilqr.py:100-265).
  c5  --batch tuning candidates x 200-step closed loop, scores all-gathered (BASELINE config 5).
Inputs are resident in HBM when the timed region starts.  For N > 1 every rank (one process per
GPU) works on its own independent problems / candidate shard -- a single solve does not shard
 -- so scaling is weak and there is no collective on the data path except
config 5's one score all-gather; ranks meet at the barriers that bracket the timed region and at
the final max-over-ranks reduction of the elapsed time.

`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (torch.distributed.run);
under a launcher (WORLD_SIZE set) it is one of the ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # MI355X dense MFMA peaks for the arithmetic type used
EVENT_STRIDE = 8                             # kernel events around every 8th timed MPPI solve (roofline.kernel_ms)
PROFILE_TAG = "r06"                          # profiles/<tag>_hbm_traffic.json feeds roofline.traffic


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c4", "c5", "arx"],
                    help="c3 (default, headline): HalfCheetah MPPI 4096x30; c2: Pendulum MPPI; "
                         "c1: CartPole SINDy MPPI 256x20 (scalar kernels); "
                         "arx: MPPI on a linear ARX model (SURVEY 8 f3); "
                         "c4: HalfCheetah iLQR H=50, --batch problems per step; c5: --batch tuning "
                         "candidates x 200-step closed loop per step")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"])
    ap.add_argument("--noise", default="device", choices=["device", "resident"],
                    help="device: fresh Philox noise generated on the GPU inside every step; "
                         "resident: one numpy-drawn noise set uploaded before timing and reused")
    ap.add_argument("--batch", type=int, default=0,
                    help="independent problems per step per GPU (default: 1 solve; c4 256; c5 64)")
    ap.add_argument("--tile-rows", type=int, default=32, choices=[16, 32, 64],
                    help="c5: samples per rollout workgroup of the candidate evaluator")
    ap.add_argument("--preheat", type=float, default=1.0,
                    help="seconds of untimed solves before the warm-up steps, so that short runs "
                         "(--steps 20) are timed at the settled clock like long ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the f32 fast-mode side report")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="time budget of EACH cpu_baseline leg (all cores, one thread)")
    ap.add_argument("--repeats", type=int, default=7,
                    help="default line: the timed window of --steps solves is repeated this many times "
                         "(value = the first window, as the contract says; repeat_windows = median / min / max)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------
# CPU baseline: the oracle (kind "port") on the host's cores -- BASELINE.md section 3
# ----------------------------------------------------------------------------------------------
def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed_legs(run_once, budget_s, min_all=3, target=20, only=None):
    """Two legs -- all host cores, then one thread -- each: one untimed call, then consecutive timed
    calls until `target` of them or `budget_s` seconds (at least min_all / 1).  Per-call times ->
    median, p10, p90.  (torch's intra-op pool follows the same limit.)"""
    import torch
    from threadpoolctl import threadpool_limits
    legs = {}
    torch_default = torch.get_num_threads()
    for name, limit, lo in (("all_cores", None, min_all), ("one_thread", 1, 1)):
        if only and name not in only:
            continue
        torch.set_num_threads(limit if limit else torch_default)
        with threadpool_limits(limits=limit if limit else os.cpu_count()):
            t0 = time.perf_counter()
            run_once()
            first = time.perf_counter() - t0
            times, t_start = [], time.perf_counter()
            if first > budget_s:              # one call already exceeds the leg's budget: it is the sample
                times, lo = [first], 0
            while len(times) < target and (len(times) < lo or time.perf_counter() - t_start < budget_s) and lo > 0:
                t0 = time.perf_counter()
                run_once()
                times.append(time.perf_counter() - t0)
        t = np.array(times)
        legs[name] = {"value": float(1.0 / np.median(t)), "unit": "solves/s", "solves": len(times),
                      "median_s": float(np.median(t)), "p10_s": float(np.percentile(t, 10)),
                      "p90_s": float(np.percentile(t, 90)), "threads": limit or os.cpu_count()}
    torch.set_num_threads(torch_default)
    return legs


def _baseline_record(legs, sample, torch_legs=None):
    # the FASTEST leg is the baseline (on a many-core host the tiny per-step GEMMs of this path run
    # slower on all cores than on one); all legs are reported in full.  torch_legs: the same workload with
    # the model evaluated the way the reference does it (oracle.mlp.MLPOracleTorch: torch f64 nn.Linear on
    # the CPU, numpy <-> torch copies and per-column normalisation loops per call, mlp.py:20-30,219-236) --
    # SURVEY.md section 8d's protocol; the numpy restatement is the faster one, so the ratio is conservative
    every = dict(legs)
    if torch_legs:
        every.update({"torch_" + k: v for k, v in torch_legs.items()})
    best = max(every.values(), key=lambda leg: leg["value"])
    if "all_cores" not in legs:       # trimmed record of a sub-record: the one-thread leg only
        one = legs["one_thread"]
        return {"value": one["value"], "unit": "solves/s", "cores": 1, "kind": "port", "cpu": cpu_model_name(),
                "host_cores": os.cpu_count(),
                "sample": "%s; one-thread leg only, %d solves (trimmed: the full two-leg record is "
                          "`bench.py --workload ...`'s); value = 1 / median solve time" % (sample, one["solves"]),
                "one_thread": one}
    rec = {"value": best["value"], "unit": "solves/s", "cores": best["threads"], "kind": "port",
           "cpu": cpu_model_name(), "host_cores": os.cpu_count(),
           "sample": "%s; all-core leg %d solves, one-thread leg %d solves; value = 1 / median solve "
                     "time of the fastest leg (%d thread(s))"
                     % (sample, legs["all_cores"]["solves"], legs["one_thread"]["solves"], best["threads"]),
           "all_cores": legs["all_cores"], "one_thread": legs["one_thread"]}
    if torch_legs:
        rec["reference_call_structure"] = dict(
            note="model evaluated as the reference does (torch f64 nn.Linear on the CPU, numpy <-> torch copies, "
                 "per-column normalisation loops: mlp.py:20-30, 219-236)", **torch_legs)
    return rec


def cpu_baseline_mppi(workload, spec, budget_s):
    """MPPI: the oracle in strict_reference mode (per-step pred_batch + the reference's
    per-particle Python cost loop, mppi.py:73-78), consecutive run() calls feeding back the
    controller state."""
    from oracle.costs import QuadCostOracle
    from oracle.mlp import MLPOracle, make_params
    from oracle.mppi import MPPIOracle
    from autompc_amd import System
    nx, nu = spec["nx"], spec["nu"]
    no = spec.get("obs", nx)
    system = System(["x%d" % i for i in range(no)], ["u%d" % i for i in range(nu)], dt=0.05)
    if "sindy" in spec:
        from oracle.sindy import SINDyOracle
        sd = spec["sindy"]
        model = SINDyOracle(system, sd["Xi"], trig_freq=sd["trig_freq"], trig_interaction=sd["trig_interaction"],
                            poly_degree=sd["poly_degree"], time_mode="discrete")
        x0 = np.array([0.0, 0.2, 0.0, 0.0])
        cs = np.concatenate([x0, np.zeros(nu)])
    elif "linear" in spec:
        from oracle.linear import ARXOracle
        model = ARXOracle(system, spec["history"], *spec["linear"])
        x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=no)
        cs = np.concatenate([model.state_from_first_obs(x0), np.zeros(nu)])
    else:
        p = spec["params"]
        model = MLPOracle(system, make_params(p["weights"], p["biases"], "relu", p["xu_means"],
                                              p["xu_std"], p["dy_means"], p["dy_std"]))
        x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=nx)
        cs = np.concatenate([x0, np.zeros(nu)])
    cost = QuadCostOracle(np.eye(no), 0.01 * np.eye(nu), np.eye(no), np.zeros(no))
    if "sindy" in spec:
        cost = QuadCostOracle(np.diag([1.0, 10.0, 0.1, 0.1]), 0.01 * np.eye(nu), np.eye(no), np.zeros(no))
    bnd = np.tile([-spec["bound"], spec["bound"]], (nu, 1))
    np.random.seed(0)
    ctl = MPPIOracle(model, cost, bnd, horizon=spec["horizon"],
                     num_path=spec["num_path"], sigma=1.0, lmda=1.0, strict_reference=True)
    state = {"cs": cs}

    def run_once():
        _, state["cs"] = ctl.run(state["cs"], x0)
    legs = _timed_legs(run_once, budget_s)
    torch_legs = None
    if "sindy" not in spec and "linear" not in spec:
        from oracle.mlp import MLPOracleTorch
        np.random.seed(0)
        ctl = MPPIOracle(MLPOracleTorch(system, model.params), cost, bnd, horizon=spec["horizon"],
                         num_path=spec["num_path"], sigma=1.0, lmda=1.0, strict_reference=True)
        state["cs"] = cs
        # (one thread only: on all 256 hardware threads torch's tiny f64 GEMMs take 24 s per c3 solve --
        #  0.041 solves/s, measured once in round 4 -- which would be most of this bench's run time)
        torch_legs = _timed_legs(run_once, budget_s / 2, target=10, only=("one_thread",))
    return _baseline_record(legs,
                            "consecutive %s MPPI solves feeding back the controller state (oracle: numpy f64 "
                            "pred_batch per step + the reference's per-particle Python cost loop)" % workload,
                            torch_legs)


def cpu_baseline_ilqr(system, spec, x0s, budget_s, bounded=False, only=None):
    from oracle.costs import QuadCostOracle
    from oracle.ilqr import ILQROracle
    from oracle.mlp import MLPOracle, make_params
    nx, nu = spec["nx"], spec["nu"]
    p = spec["params"]
    om = MLPOracle(system, make_params(p["weights"], p["biases"], "relu", p["xu_means"],
                                       p["xu_std"], p["dy_means"], p["dy_std"]))
    orc = ILQROracle(om, QuadCostOracle(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx)),
                     system.dt, 50, ubounds=(np.full(nu, -0.25), np.full(nu, 0.25)) if bounded else None)
    k = {"i": 0}

    def run_once():
        orc.solve(x0s[k["i"] % len(x0s)], np.zeros((50, nu)))
        k["i"] += 1
    return _baseline_record(_timed_legs(run_once, budget_s, only=only),
                            "HalfCheetah iLQR H=50 solves from the bench's initial states%s (oracle: numpy f64)"
                            % (", controls clipped to +-0.25" if bounded else ""))


def cpu_baseline_c5(system, spec, cands, budget_s, n_ctl=16, only=None):
    """One control step of one candidate's closed loop = one MPPI solve + one surrogate step: the
    unit the c5 value counts.  Same structure as the c3 baseline: the oracle in strict_reference
    mode (per-step pred_batch + the reference's per-particle Python cost loop, mppi.py:73-78),
    steady state -- the first `n_ctl` candidates' controllers are built and take one closed-loop
    step before timing; every timed call advances the next controller's closed loop by one step."""
    from oracle.costs import QuadCostOracle
    from oracle.mlp import MLPOracle, make_params
    from oracle.mppi import MPPIOracle
    nx, nu = spec["nx"], spec["nu"]
    p = spec["params"]
    om = MLPOracle(system, make_params(p["weights"], p["biases"], "relu", p["xu_means"],
                                       p["xu_std"], p["dy_means"], p["dy_std"]))
    x0 = np.random.default_rng(0).uniform(-0.1, 0.1, size=nx)
    loops = []
    np.random.seed(0)
    for c in cands[:n_ctl]:
        ctl = MPPIOracle(om, QuadCostOracle(np.diag(c["Q"]), np.diag(c["R"]), np.diag(c["F"]), np.zeros(nx)),
                         np.tile([-1.0, 1.0], (nu, 1)), horizon=c["horizon"], num_path=c["num_path"],
                         sigma=c["sigma"], lmda=c["lmda"], strict_reference=True)
        loops.append({"ctl": ctl, "x": x0.copy(), "cs": np.concatenate([x0, np.zeros(nu)])})
    state = {"i": 0}

    def advance(lp):
        u, lp["cs"] = lp["ctl"].run(lp["cs"], lp["x"])
        lp["x"] = om.pred(lp["x"], u)
    for lp in loops:
        advance(lp)

    def run_once():
        advance(loops[state["i"] % len(loops)])
        state["i"] += 1
    return _baseline_record(_timed_legs(run_once, budget_s, target=4 * len(loops), only=only),
                            "steady-state closed-loop control steps (MPPI solve + surrogate step) of the bench's "
                            "first %d candidates in turn (oracle: numpy f64 pred_batch per step + the reference's "
                            "per-particle Python cost loop)" % len(loops))


# ----------------------------------------------------------------------------------------------
def measured_traffic(args, batch, kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/<tag>_hbm_traffic.json, written by tools/summarize_profiles.py from separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command).  Counters cannot
    be collected from inside the timed process, so the figure is reported only for the
    configuration it was measured on."""
    path = os.path.join(ROOT, "profiles", "%s_hbm_traffic.json" % PROFILE_TAG)
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        key = "%s_%s_b%d" % (args.workload, args.precision, batch)
        entry = d.get(key)
        if entry is None:
            return None
        for name, v in entry["kernels"].items():
            if kernel_substr in name:
                return {"bytes": v["bytes"], "source": "profiles/%s_hbm_traffic.json[%s] (%s)"
                                                       % (PROFILE_TAG, key, entry["method"])}
    except (OSError, ValueError, KeyError):
        pass
    return None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks here, one
    process per GPU, exactly as the driver's own command line does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
           str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Ranks:
    """Rank plumbing shared by all workloads: barrier + synchronize brackets, max over ranks."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # Plumbing test hooks (tools/gpu_round.sh): run a 2-rank job on a 1-GPU box by mapping every
        # rank to one device and using gloo for the barriers.  Never set by the driver.
        if "AMPC_BENCH_FORCE_DEVICE" in os.environ:
            self.local_rank = int(os.environ["AMPC_BENCH_FORCE_DEVICE"])
        backend = os.environ.get("AMPC_BENCH_BACKEND", "nccl")
        if self.local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d needs GPU %d but only %d device(s) are visible "
                             "(--gpus %d)" % (self.rank, self.local_rank, torch.cuda.device_count(), args.gpus))
        torch.cuda.set_device(self.local_rank)
        self.pinned = None
        if self.world > 1:
            # this rank's host threads next to its GPU: the CPUs of the GPU's NUMA node, divided among the ranks
            # that share the node; pools capped at the share (autompc_amd/tuning/hostpin.py; AMPC_PIN=0 disables)
            from autompc_amd.tuning.hostpin import pin_rank
            self.pinned = pin_rank(int(os.environ.get("LOCAL_RANK", "0")),
                                   int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world))))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend)

    def sync_all(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, elapsed):
        if self.world == 1:
            return elapsed
        t = self.torch.tensor([elapsed], dtype=self.torch.float64,
                              device="cuda" if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed_loop(R, step, steps, warmup, preheat_s, before_timed=None):
    """preheat_s seconds of untimed steps (clock settling), W untimed warm-up steps, then EXACTLY
    `steps` timed steps bracketed by barrier + synchronize; returns (elapsed max over ranks,
    number of pre-heat steps)."""
    n_pre, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < preheat_s:
        step(-1 - n_pre)
        n_pre += 1
        if n_pre % 16 == 0:
            R.torch.cuda.synchronize()
    for i in range(warmup):
        step(i)
    if before_timed is not None:
        before_timed()
    R.sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    R.sync_all()
    return R.max_over_ranks(time.perf_counter() - t0), n_pre


# ----------------------------------------------------------------------------------------------
# c4 / c5
# ----------------------------------------------------------------------------------------------
def secondary_workload(args, R, emit=True):
    """c4 / c5 as the workload of the line (emit) or as a sub-record of the default line (returned)."""
    from autompc_amd import _lib
    from autompc_amd.synthetic import make_workload
    system, task, model, spec = make_workload("c3", precision=args.precision, device=R.local_rank)
    nx, nu = spec["nx"], spec["nu"]
    world, rank = R.world, R.rank
    # c4: --batch B = B slots with 4 B problems streamed through them (continuous batching); default: 1024 problems
    # admitted at once (1024 slots, no drain -- round 5: the kernels of an iteration take the slots with work first)
    B = args.batch if args.batch > 0 else (64 if args.workload == "c5" else 1024)
    mlp_macs = sum(a * b for a, b in zip([nx + nu] + spec["hidden"], spec["hidden"] + [nx]))
    extra = {}
    steps, warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
    peak = PEAK_TFLOPS[args.precision]
    roof = None

    if args.workload == "c4":
        # Headline c4 = the CONVERGING problem set (controls clipped to +-0.25: the reference's bounded
        # golden problem, tests/golden/ilqr_hc6_relu_bounded.npz; clipping in the forward pass,
        # ilqr.py:62-64, 203-204): P independent problems streamed through B slots with continuous
        # batching (ampc_ilqr_solve_queue) -- a slot whose problem has converged takes the next one at
        # the following iteration boundary, on the device.  One step = P complete solves.
        Q, Rm, F = task.get_cost().get_cost_matrices()
        P = 4 * B if args.batch > 0 else B
        rng = np.random.default_rng(rank)
        x0 = rng.uniform(-0.1, 0.1, size=(max(P, 4096), nx))
        stream = R.torch.cuda.current_stream().cuda_stream

        def make(bounded, slots=0, bound=0.25):
            hh = _lib.Handle(R.local_rank, args.precision, stream=stream)
            model.stage_into(hh)
            hh.set_quad_costs(Q, Rm, F, task.get_cost().get_goal())
            if bounded:
                hh.set_ctrl_bounds(np.full(nu, -bound), np.full(nu, bound))
            return hh, _lib.IlqrPlan(hh, slots or B, 50, system.dt, clip_to_bounds=bounded)
        h, plan = make(True)
        last = {}

        def step(i):
            last["out"] = plan.solve_queue(x0[:P], max_iter=50, gains=False, trajectories=False)
            if i >= 0:
                last.setdefault("rows", []).append(plan.stats()["candidate_rows"])
                last.setdefault("launched", []).append(plan.stats()["iterations"])
        label = ("c4: HalfCheetah MLP 2x256, iLQR horizon 50, controls clipped to +-0.25 (converging set): %d "
                 "independent problems per step per GPU " % P +
                 ("streamed through %d slots (continuous batching)" % B if P > B else
                  "admitted at once (%d slots; a slot is idle once its problem has converged)" % B))
        unit_per_step = P
        metric, unit = "MPC solves/sec (iLQR, full compute_ilqr_default per solve)", "solves/s"
        elapsed, n_pre = timed_loop(R, step, steps, warm, min(args.preheat, 0.3))
        ob = last["out"]
        # per-kernel times: ONE more step of the same problems with HIP events around the launches of every iteration
        # (their work differs: all of them are bracketed) -- five event records per iteration cost 1-2 % of the step,
        # so they stay out of the timed region; the kernel durations are the same
        plan.set_timing(True)
        plan.solve_queue(x0[:P], max_iter=50, gains=False, trajectories=False)
        kt = plan.timing()
        plan.set_timing(False)
        kt_steps = 1
        sub = {}
        if not args.no_extras:
            def once(fn):
                fn()
                R.sync_all()
                t0 = time.perf_counter()
                o = fn()
                R.sync_all()
                return o, R.max_over_ranks(time.perf_counter() - t0)
            B0 = 256                       # one slot per CU: the configuration of rounds 3-4
            if B != B0:
                h0, plan0 = make(True, B0)
                o, e = once(lambda: plan0.solve_queue(x0[:P], max_iter=50, gains=False, trajectories=False))
                sub["slots_256"] = {"workload": "the same %d problems streamed through %d slots (rounds 3-4)" % (P, B0),
                                    "value": world * P / e, "unit": "solves/s", "ms": 1e3 * e,
                                    "iterations_launched": plan0.stats()["iterations"]}
            else:
                h0, plan0 = h, plan
            if B < 4096 and not args.batch:
                h4, plan4 = make(True, 4096)
                o, e = once(lambda: plan4.solve_queue(x0[:4096], max_iter=50, gains=False, trajectories=False))
                sub["all_at_once_4096"] = {"workload": "4096 problems admitted at once (4096 slots)",
                                           "value": world * 4096 / e, "unit": "solves/s", "ms": 1e3 * e,
                                           "iterations_launched": plan4.stats()["iterations"]}
                plan4.close()
                h4.close()
            # a long stream: the drain of the last (non-converging, 50-iteration) problems amortised
            o, e = once(lambda: plan.solve_queue(x0[:4096], max_iter=50, gains=False, trajectories=False))
            sub["stream_4096"] = {"workload": "the same set, 4096 problems through %d slots" % B,
                                  "value": world * 4096 / e, "unit": "solves/s", "ms": 1e3 * e,
                                  "converged_fraction": float(o["converged"].mean()),
                                  "mean_iterations_per_solve": float(o["iters"].mean())}
            # two independent queues of B slots each (own handle / stream / host thread): the kernels of one
            # fill the compute units the other's lock-step launches leave idle (decided line searches,
            # retired slots)
            import threading
            h2 = _lib.Handle(R.local_rank, args.precision)          # (a stream of its own)
            model.stage_into(h2)
            h2.set_quad_costs(Q, Rm, F, task.get_cost().get_goal())
            h2.set_ctrl_bounds(np.full(nu, -0.25), np.full(nu, 0.25))
            plan2 = _lib.IlqrPlan(h2, B, 50, system.dt, clip_to_bounds=True)

            def two_queues():
                res = [None, None]

                def run(k, pl, xs):
                    res[k] = pl.solve_queue(xs, max_iter=50, gains=False, trajectories=False)
                th = [threading.Thread(target=run, args=(0, plan, x0[:2048])),
                      threading.Thread(target=run, args=(1, plan2, x0[2048:4096]))]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                h2.synchronize()
                return {k: np.concatenate([r[k] for r in res]) for k in ("converged", "iters")}
            o, e = once(two_queues)
            sub["two_queues_4096"] = {"workload": "4096 problems through two queues of %d slots on two streams" % B,
                                      "value": world * 4096 / e, "unit": "solves/s", "ms": 1e3 * e,
                                      "converged_fraction": float(o["converged"].mean())}
            plan2.close()
            h2.close()
            # the same P problems as lock-step batches of B (ampc_ilqr_solve: a batch lasts as long as
            # its slowest problem) -- what round 3 measured

            def batches():
                res = [plan0.solve(x0[lo:lo + B0], np.zeros((B0, 50, nu)), max_iter=50) for lo in range(0, P, B0)]
                return {k: np.concatenate([r[k] for r in res]) for k in ("converged", "iters")}
            o, e = once(batches)
            sub["lockstep_batches"] = {"workload": "the same %d problems as %d lock-step batches of %d" % (P, P // B0, B0),
                                       "value": world * P / e, "unit": "solves/s", "ms": 1e3 * e,
                                       "converged_fraction": float(o["converged"].mean())}
            if plan0 is not plan:
                plan0.close()
                h0.close()
            # SURVEY 8(d)'s protocol for config 4: "iLQR H=50 on C3's model, unbounded and bounded (+-1) variants,
            # solves/s AND iterations/solve" -- the same P problems through the same queue, next to the +-0.25
            # converging set of the headline.  slot-iterations/s = solves/s x iterations/solve is the rate that
            # does not depend on how soon a problem set converges.  with_outputs: Ks / ks / states / ctrls of
            # every problem downloaded as well (the headline step leaves them on the device).
            proto = {}
            for tag, bounded, bound in (("unbounded", False, 0.0), ("bounded_1.0", True, 1.0)):
                hp, pp = make(bounded, B, bound)
                o, e = once(lambda: pp.solve_queue(x0[:P], max_iter=50, gains=False, trajectories=False))
                its = float(o["iters"].mean())
                proto[tag] = {"workload": "%d problems, %s" % (P, "controls clipped to +-%g" % bound if bounded else "no control bounds"),
                              "value": world * P / e, "unit": "solves/s", "ms": 1e3 * e,
                              "mean_iterations_per_solve": its, "converged_fraction": float(o["converged"].mean()),
                              "slot_iterations_per_s": world * P * its / e}
                pp.close()
                hp.close()
            o, e = once(lambda: plan.solve_queue(x0[:P], max_iter=50, gains=True, trajectories=True))
            its = float(o["iters"].mean())
            proto["bounded_0.25_with_outputs"] = {
                "workload": "the headline set (+-0.25) with states / ctrls / Ks / ks of all %d problems downloaded "
                            "(%.1f MB)" % (P, P * 8 * (51 * nx + 50 * nu + 50 * nu * nx + 50 * nu) / 1e6),
                "value": world * P / e, "unit": "solves/s", "ms": 1e3 * e, "mean_iterations_per_solve": its,
                "slot_iterations_per_s": world * P * its / e}
            sub["protocol_variants"] = proto
            # the unbounded problems of rounds 1-3: none converges within the reference's 50 iterations
            hu, pu = make(False, B0)
            o, e = once(lambda: pu.solve(x0[:B0], np.zeros((B0, 50, nu)), max_iter=50))
            sub["capped_variant"] = {"workload": "%d UNBOUNDED problems (never converge: 50-iteration capped solves)" % B0,
                                     "value": world * B0 / e, "unit": "solves/s", "ms": 1e3 * e,
                                     "converged_fraction": float(o["converged"].mean()),
                                     "mean_iterations_per_solve": float(o["iters"].mean())}
            pu.close()
            hu.close()
        if rank == 0:
            # work of one solve: per iteration the Jacobian chain over H rows and the forward
            # pass of the accepted trajectory; the line search as EXECUTED -- candidate rows rolled out
            # (four per pass, or all ten in one twelve-row pass; the reference rolls out all ten step
            # sizes every iteration, ilqr.py:196-205, the same arithmetic per row); + the rollout of the guess
            it = float(ob["iters"].mean())
            ls_rows = float(np.mean(last["rows"])) / P
            hid = sum(a * b for a, b in zip(spec["hidden"], spec["hidden"][1:]))
            jac = 50 * 2 * nx * (hid + spec["hidden"][0] * (nx + nu))
            row = 50 * 2 * mlp_macs
            per_solve = (it + 1) * (jac + row) + ls_rows * row
            extra["problems_per_step"] = P
            extra["slots"] = B
            extra["mean_iterations_per_solve"] = it
            extra["converged_fraction"] = float(ob["converged"].mean())
            extra["slot_iterations_per_s"] = world * steps * P * it / elapsed
            extra["outputs_left_on_device"] = ("the timed step downloads objective / converged / iterations per problem; "
                                               "states, ctrls, Ks, ks stay on the device (protocol_variants."
                                               "bounded_0.25_with_outputs downloads them too)")
            extra["iteration_cap"] = 50
            extra["iterations_launched_per_step"] = float(np.mean(last["launched"]))
            extra["ideal_iterations_per_step"] = P * (it + 1) / B
            extra["mean_line_search_rows_per_iteration"] = ls_rows / it
            extra["algorithmic_tflops"] = world * steps * P * per_solve / elapsed / 1e12
            extra["reference_work_tflops"] = world * steps * P * ((it + 1) * (jac + row) + it * 10 * row) / elapsed / 1e12
            extra.update(sub)
            if kt and kt.get("launches"):
                # per launch over the B slots; the busy fraction of the slots varies (drain), so the
                # per-launch work is averaged over the launches of the timed steps
                n_l = max(kt["launches"], 1)
                jac_fl = kt_steps * P * (it + 1) * jac / n_l
                ls_fl = kt_steps * P * ls_rows * row / n_l
                # (line search: ilqr_lsw_kernel -- all step sizes in one twelve-row pass -- once some search of
                #  the launch needs a third four-row pass, else ilqr_ls4_kernel; chosen by the plan per poll)
                cand = {"jacobian": (jac_fl, "mlp_jacobian_kernel"), "iter": (ls_fl, "ilqr_lsw_kernel")}
                dom = max(cand, key=lambda k: kt.get(k + "_ms", 0.0))
                fl, kname = cand[dom]
                ach = fl / (kt[dom + "_ms"] * 1e-3) / 1e12
                tr = measured_traffic(args, B, kname)
                roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                        "kernel": kname, "kernel_ms": kt[dom + "_ms"],
                        "per_iteration_kernel_ms": {k: kt.get(k + "_ms") for k in
                                                    ("riccati", "iter", "forward", "jacobian")},
                        "iterations_timed": kt["launches"], "algorithmic_flops_per_launch": fl,
                        "other_kernel": {k: {"achieved": cand[k][0] / (kt[k + "_ms"] * 1e-3) / 1e12,
                                             "frac": cand[k][0] / (kt[k + "_ms"] * 1e-3) / 1e12 / peak}
                                         for k in cand if k != dom},
                        "events": "one separate step after the timed ones, every iteration bracketed",
                        "note": "the kernel with the largest share of an iteration; one launch covers the %d "
                                "slots (work averaged over the timed launches, drain included); whole-solve "
                                "rate in algorithmic_tflops" % B}
    else:
        from autompc_amd.tuning import CandidateEvaluator, evaluate_sharded, random_candidates
        task.set_num_steps(200)
        from autompc_amd.tuning.batch_eval import default_episode_controls
        n_ctl = default_episode_controls(task)     # eval_cfg's episode: 200 rows = 199 control steps
        # BASELINE config 5: one candidate list for the whole job, shards of ~B per GPU balanced by work,
        # randomness keyed by the global candidate index (scores independent of the world size),
        # scores exchanged with one all-gather (RCCL over xGMI under the nccl backend)
        cands = random_candidates(system, B * world, seed=0)
        ev = CandidateEvaluator(system, task, model, precision=args.precision, device=R.local_rank,
                                tile_rows=args.tile_rows)
        last = {}

        def step(i):
            scores = evaluate_sharded(
                lambda shard, lo: ev.evaluate(shard, seed=max(i, 0), index_offset=lo, timing=last),
                cands, weights="auto")      # shards balanced by num_path x horizon (the gather is max-over-ranks bound)
            if not np.all(np.isfinite(scores)) or scores.shape[0] != B * world:
                raise RuntimeError("candidate scores incomplete")
        label = ("c5: %d tuning candidates (MPPI horizon/sigma/lmda/num_path + QuadCost weights from "
                 "the reference's config ranges) x one eval_cfg episode (task.set_num_steps(200): 200 rows = "
                 "%d closed-loop control steps, pipeline_tuner.py:222-231) per step per GPU" % (B, n_ctl))
        unit_per_step = B * n_ctl
        extra["control_steps_per_episode"] = n_ctl
        metric, unit = "MPC solves/sec (MPPI inside the batched closed-loop candidate evaluator)", "solves/s"
        elapsed, n_pre = timed_loop(R, step, steps, warm, 0.0)
        kt = last.get("timing")
        if rank == 0:
            from autompc_amd.tuning import balanced_shards, candidate_work
            mine = balanced_shards([candidate_work(c) for c in cands], world)[0]      # rank 0's shard
            extra["candidates_on_rank0"] = int(len(mine))
            per_ctrl_step = sum(cands[int(k)]["num_path"] * cands[int(k)]["horizon"] for k in mine)
            flops = per_ctrl_step * (2 * mlp_macs + 2 * (nx * nx + nx) + 2 * nu * nu + 2 * nu)
            all_steps = sum(c["num_path"] * c["horizon"] for c in cands) * \
                (2 * mlp_macs + 2 * (nx * nx + nx) + 2 * nu * nu + 2 * nu)
            extra["algorithmic_tflops"] = steps * n_ctl * all_steps / elapsed / 1e12
            if kt and kt.get("count"):
                ach = flops / (kt["rollout_ms"] * 1e-3) / 1e12
                tr = measured_traffic(args, B, "mppi_rollout_kernel")
                roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                        "kernel": "mppi_rollout_kernel", "kernel_ms": kt["rollout_ms"],
                        "update_kernel_ms": kt["update_ms"], "launches_timed": kt["count"], "event_stride": 1,
                        "algorithmic_flops_per_launch": flops,
                        "note": "one launch = one control step of all %d candidates of rank 0" % B}
    if rank == 0:
        extra["mfma_peak_tflops"] = peak
        out = {"metric": metric, "value": world * steps * unit_per_step / elapsed, "unit": unit,
               "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * elapsed / steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.precision, "data": "synthetic", "preheat_steps": n_pre,
               "config": {"workload": label, "parallelism": "independent problems per GPU (dp%d)" % world},
               "roofline": roof, **extra}
        trimmed = getattr(args, "trimmed_cpu_seconds", 0.0)     # (sub-record of the default line)
        if (not args.no_cpu_baseline or trimmed > 0) and world == 1:
            only = ("one_thread",) if trimmed > 0 else None
            budget = trimmed if trimmed > 0 else args.cpu_seconds
            if args.workload == "c4":
                out["cpu_baseline"] = cpu_baseline_ilqr(system, spec, x0, budget, bounded=True, only=only)
            else:
                out["cpu_baseline"] = cpu_baseline_c5(system, spec, cands, budget, n_ctl=4 if trimmed > 0 else 16, only=only)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if emit:
            print(json.dumps(out))
        return out
    return None


def ilqr_eval_record(args, R, n_cand=64, n_rows=50):
    """The tuner's OTHER controller on the same surrogate: iLQR candidates from the reference's ranges
    (IterativeLQRFactory horizon 5..25, control/ilqr.py:31-41; QuadCostFactory gains) x one eval_cfg episode
    each (pipeline_tuner.py:222-231), device resident, all horizons through one plan
    (ampc_ilqr_closed_loop_var).  One solve = one full compute_ilqr_default from a zero guess."""
    from autompc_amd.synthetic import make_workload
    from autompc_amd.tuning import IlqrCandidateEvaluator, random_ilqr_candidates
    system, task, model, spec = make_workload("c3", precision="f64", device=R.local_rank)
    task.set_num_steps(n_rows)
    cands = random_ilqr_candidates(system, n_cand, seed=0)
    for c in cands:                    # gains 0.18 .. 10 (the ranges' fourth root): O(1) stage costs on this surrogate
        c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25
    ev = IlqrCandidateEvaluator(system, task, model, device=R.local_rank)
    ev.evaluate(cands[:8])
    R.sync_all()
    t0 = time.perf_counter()
    scores = ev.evaluate(cands)
    elapsed = time.perf_counter() - t0
    n_ctl = n_rows - 1
    return {"workload": "%d iLQR candidates (horizons %d..%d, QuadCost gains) x one eval_cfg episode of %d rows = %d "
                        "control steps on the HalfCheetah surrogate; one plan for every horizon, episodes device resident"
                        % (n_cand, min(c["horizon"] for c in cands), max(c["horizon"] for c in cands), n_rows, n_ctl),
            "value": n_cand * n_ctl / elapsed, "unit": "solves/s", "elapsed_s": elapsed,
            "mean_iterations_per_solve": float(ev.last_iterations.sum() / (n_cand * n_ctl)),
            "scores_finite": int(np.isfinite(scores).sum())}


def model_axis_record(args, R, n_cand=64, n_models=8, n_rows=200, n_traj=40, traj_rows=201, epochs=50,
                      seq_epochs=2):
    """The tuner's MODEL axis: eval_cfg fits a model per configuration (pipeline.py:138-145 ->
    MLP.train, sysid/mlp.py:177-217: Adam, SmoothL1, 50 epochs of 64-row mini-batches) before it simulates.
    `n_models` distinct 2x256 configurations (own seeds / learning rates) fitted on one trajectory set
    (`n_traj` surrogate rollouts under random controls), then `n_cand` MPPI candidates spread over them, one
    eval_cfg episode each.  fit_s = the lockstep PyTorch-ROCm fit of all models (HIP-graph captured, capture
    time included); sequential_fit_s = the same models one after the other with nn.Linear + torch.optim.Adam on
    the GPU, extrapolated from `seq_epochs` epochs of one model; stage_s = ampc_set_mlp_dev for all models;
    eval_s = the batched closed-loop evaluation."""
    import torch
    from autompc_amd import MLP, zeros
    from autompc_amd.synthetic import make_workload
    from autompc_amd.sysid import mlp_fit
    from autompc_amd.tuning import CandidateEvaluator, random_candidates
    system, task, model, spec = make_workload("c3", precision="f64", device=R.local_rank)
    task.set_num_steps(n_rows)
    rng = np.random.default_rng(0)
    nx, nu = system.obs_dim, system.ctrl_dim
    X = rng.uniform(-0.1, 0.1, size=(n_traj, nx))
    trajs = [zeros(system, traj_rows) for _ in range(n_traj)]
    for t in range(traj_rows):                       # the surrogate under random bounded controls
        U = rng.uniform(-1.0, 1.0, size=(n_traj, nu))
        for k in range(n_traj):
            trajs[k].obs[t], trajs[k].ctrls[t] = X[k], U[k]
        X = model.pred_batch(X, U)
    models = [MLP(system, n_hidden_layers=2, hidden_size=256, nonlintype="relu", n_train_iters=epochs, n_batch=64,
                  lr=1e-3 * (1 + 0.25 * k), seed=k, device=R.local_rank) for k in range(n_models)]
    XU, dY, xm, xs, dm, ds = mlp_fit.training_arrays(trajs)
    feed, target = [torch.from_numpy(v).cuda(R.local_rank) for v in mlp_fit.normalised(XU, dY, xm, xs, dm, ds)]
    dims = [nx + nu, 256, 256, nx]
    dev = "cuda:%d" % R.local_rank
    mlp_fit.fit_reference_style(dims, "relu", feed[:256], target[:256], 1, 64, 1e-3, 0, device=dev)   # library warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mlp_fit.fit_reference_style(dims, "relu", feed, target, seq_epochs, 64, 1e-3, 0, device=dev)
    torch.cuda.synchronize()
    seq_one = (time.perf_counter() - t0) * epochs / seq_epochs
    info = mlp_fit.fit_mlps(models, trajs, device=dev)
    t0 = time.perf_counter()
    for m in models:
        m._dev()
    stage_s = time.perf_counter() - t0
    staged_from_device = all(m._weights is None for m in models)
    # held-out one-step error of the fitted models against the surrogate that generated the data
    S, U = rng.uniform(-0.3, 0.3, size=(512, nx)), rng.uniform(-1.0, 1.0, size=(512, nu))
    truth = model.pred_batch(S, U)
    rel = [float(np.linalg.norm(m.pred_batch(S, U) - truth) / np.linalg.norm(truth - S)) for m in models]
    cands = random_candidates(system, n_cand, seed=0)
    for i, c in enumerate(cands):
        c["model"] = models[i % n_models]
    ev = CandidateEvaluator(system, task, model, device=R.local_rank)
    ev.evaluate(cands[:8], seed=1)
    R.sync_all()
    t0 = time.perf_counter()
    scores = ev.evaluate(cands, seed=1)
    eval_s = time.perf_counter() - t0
    steps = info["steps"]
    return {"workload": "%d MPPI candidates over %d distinct 2x256 MLP configurations fitted on %d rows (%d epochs x %d "
                        "mini-batches of 64 = %d optimiser steps each), then one eval_cfg episode per candidate (%d "
                        "control steps)" % (n_cand, n_models, feed.shape[0], epochs, -(-feed.shape[0] // 64), steps, n_rows - 1),
            "fit_s": info["fit_s"], "stage_s": stage_s, "eval_s": eval_s, "fit_over_eval": info["fit_s"] / eval_s,
            "sequential_fit_s": seq_one * n_models, "lockstep_speedup": seq_one * n_models / info["fit_s"],
            "optimiser_steps_per_s": steps / info["fit_s"], "us_per_lockstep_step": 1e6 * info["fit_s"] / steps,
            "models_per_s": n_models / info["fit_s"], "staged_from_device_memory": staged_from_device,
            "heldout_rel_err_of_delta": {"min": min(rel), "max": max(rel)},
            "scores_finite": int(np.isfinite(scores).sum()),
            "note": "the fit is a chain of %d DEPENDENT optimiser steps on 64 rows (the reference's batch size): "
                    "its floor is steps x the per-step kernel chain, whatever the number of models" % steps}


def c5_sharded_record(args, R, per_gpu=64, probes=4):
    """north_star's multi-GPU path inside the default (c3) bench line of an N > 1 run: BASELINE
    config 5's candidate sharding -- per_gpu * world candidates, shards balanced by num_path x horizon, randomness keyed
    by the global candidate index, ONE all_gather_into_tensor of the scores (RCCL over xGMI under
    backend nccl) -- run once after an untimed pass.  Rank 0 then re-evaluates `probes` candidates
    spread over all shards on its own GPU: the gathered scores must be what a single rank
    computes.  Every rank takes part; rank 0 returns the record."""
    from autompc_amd.synthetic import make_workload
    from autompc_amd.tuning import CandidateEvaluator, evaluate_sharded, random_candidates
    from autompc_amd.tuning.batch_eval import balanced_shards, candidate_work, default_episode_controls
    system, task, model, spec = make_workload("c3", precision=args.precision, device=R.local_rank)
    task.set_num_steps(200)
    world = R.world
    cands = random_candidates(system, per_gpu * world, seed=0)
    ev = CandidateEvaluator(system, task, model, precision=args.precision, device=R.local_rank)

    def local(shard, lo):
        return ev.evaluate(shard, seed=0, index_offset=lo)
    evaluate_sharded(local, cands, weights="auto")       # untimed pass (plans, clocks)
    stats = {}
    R.sync_all()
    t0 = time.perf_counter()
    scores = evaluate_sharded(local, cands, stats=stats, weights="auto")
    R.sync_all()
    elapsed = R.max_over_ranks(time.perf_counter() - t0)
    if R.rank != 0:
        return None
    n = len(cands)
    owner = [set(ix.tolist()) for ix in balanced_shards([candidate_work(c) for c in cands], world)]
    idx = sorted({min(n - 1, (2 * k + 1) * n // (2 * probes)) for k in range(probes)})
    dev = 0.0
    for gi in idx:
        alone = ev.evaluate([cands[gi]], seed=0, index_offset=gi)[0]
        dev = max(dev, abs(alone - scores[gi]))
    n_ctl = default_episode_controls(task)
    return {"workload": "BASELINE config 5: %d candidates = ~%d per GPU (shards balanced by work), one eval_cfg "
                        "episode each (200 rows = %d control steps)" % (n, per_gpu, n_ctl),
            "candidates": n, "candidates_per_gpu": per_gpu, "control_steps_per_episode": n_ctl,
            "value": n * n_ctl / elapsed, "unit": "solves/s", "elapsed_s": elapsed,
            "backend": stats.get("backend"), "ranks_in_gather": stats.get("ranks_in_gather"),
            "gather_ms": stats.get("gather_ms"), "gather_device": stats.get("device"),
            "probe_candidates": idx, "probe_shards": [next(r for r in range(world) if gi in owner[r]) for gi in idx],
            "sharding": "balanced by num_path x horizon", "heaviest_rank_over_mean_work": stats.get("heaviest_over_mean"),
            "max_abs_score_deviation_vs_single_rank": dev, "scores_finite": bool(np.all(np.isfinite(scores)))}


# ----------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    R = Ranks(args)
    rank, world = R.rank, R.world

    if args.workload in ("c4", "c5"):
        secondary_workload(args, R)
        R.close()
        return

    from autompc_amd import _lib
    from autompc_amd.synthetic import make_workload
    batch = batch_default = args.batch if args.batch > 0 else 1

    def build_plan(precision, nb, workload=None):
        system, task, model, spec = make_workload(workload or args.workload, precision=precision,
                                                  device=R.local_rank, seed=0)
        stream = R.torch.cuda.current_stream().cuda_stream
        h = _lib.Handle(R.local_rank, precision, stream=stream)
        model.stage_into(h)
        Q, Rm, F = task.get_cost().get_cost_matrices()
        h.set_quad_costs(Q, Rm, F, task.get_cost().get_goal())
        bounds = task.get_ctrl_bounds()
        h.set_ctrl_bounds(bounds[:, 0], bounds[:, 1])
        N, H = spec["num_path"], spec["horizon"]
        if (workload or args.workload) == "arx":
            h.jit_wait()         # (an unregistered shape: its run-time compiled kernels, as a long-lived controller gets them)
        plan = _lib.MppiPlan(h, [N] * nb, [H] * nb, [1.0] * nb, [1.0] * nb)
        return h, plan, task, spec

    def timed_run(precision, steps, warmup, preheat_s, workload=None, windows=1, nb=None):
        """pre-heat + W untimed + K timed solves, bracketed by barrier + synchronize; max over ranks.
        windows > 1: the timed window is repeated (same plan, noise stream running on); the first
        window is the one returned as `elapsed`, all of them in `rates`.  nb: solves per launch (default:
        --batch)."""
        batch = nb if nb else batch_default
        h, plan, task, spec = build_plan(precision, batch, workload)
        nx, nu, N, H = spec["nx"], spec["nu"], spec["num_path"], spec["horizon"]
        rng = np.random.default_rng(1000 + rank)
        x0 = np.tile(spec.get("x0", task.get_init_obs()), (batch, 1))
        if "linear" not in spec:      # (an ARX state repeats the observation: leave it consistent)
            x0 = x0 + rng.uniform(-0.01, 0.01, size=(batch, nx))
        np.random.seed(rank)
        act0 = np.random.normal(size=(batch * H * nu))
        eps0 = np.random.normal(size=(batch * N * H * nu)) if args.noise == "resident" else None
        plan.upload(x0, act0, eps0)
        plan.set_outputs(keep_eps_out=False)   # nothing downloads the clipped noise here
        info = plan.info()

        def step(i):
            if args.noise == "device":
                plan.generate_eps(rank, i & 0xffffffff)
            plan.solve()
        # HIP events (on the launch stream) around the kernels of every EVENT_STRIDE-th timed solve: the three
        # event records of a solve cost ~11 us on the stream (3.7 % of a c3 solve) -- instrumentation, not work
        elapsed, n_pre = timed_loop(R, step, steps, warmup, preheat_s,
                                    before_timed=lambda: plan.set_timing(True, every=EVENT_STRIDE))
        kt = plan.timing()
        plan.set_timing(False)
        rates = [world * steps * batch / elapsed]
        for wdw in range(1, windows):
            ew, _ = timed_loop(R, lambda i: step(i + wdw * steps), steps, 0, 0.0)
            rates.append(world * steps * batch / ew)
        _, u, _, _ = plan.download(act_seq=False, u=True)
        if not np.all(np.isfinite(u)):
            raise RuntimeError("non-finite control returned by the solve")
        plan.close()
        h.close()
        spec = dict(spec, window_rates=rates)
        return elapsed, kt, info, spec, n_pre

    def mppi_record(workload, steps, warmup):
        """A compact record of another MPPI configuration (sub-record of the default line)."""
        e, k, inf, sp, _ = timed_run(args.precision, steps, warmup, 0.2, workload=workload)
        ach = inf["flops"] / (k["rollout_ms"] * 1e-3) / 1e12
        pk = PEAK_TFLOPS[args.precision]
        return {"workload": "%s: %s" % (workload, sp["label"]), "value": world * steps * batch / e, "unit": "solves/s",
                "ms_per_step": 1e3 * e / steps, "steps": steps,
                "roofline": {"bound": "mfma", "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk,
                             "traffic": None,
                             "kernel": "mppi_rollout4_kernel" if inf["samples_per_wg"] == 4 else "mppi_rollout_kernel",
                             "kernel_ms": k["rollout_ms"], "update_kernel_ms": k["update_ms"],
                             "launches_timed": k["count"], "event_stride": EVENT_STRIDE, "algorithmic_flops_per_launch": inf["flops"],
                             "workgroups": inf["workgroups"], "samples_per_workgroup": inf["samples_per_wg"],
                             "note": "latency-bound: %d workgroups of %d samples, %d dependent steps"
                                     % (inf["workgroups"], inf["samples_per_wg"], sp["horizon"])}}

    def dropin_record():
        """What a user's simulate() loop sees: Controller.run() itself, host arrays in, host control out
        (one library call + one synchronisation per control step), in the parity-graded default noise
        mode (numpy's legacy stream, generated on the device) and with device Philox noise; and
        IterativeLQR.run() (a full re-solve from a zero guess, ilqr.py:267-295) on H = 50 problems."""
        from autompc_amd import MPPI, IterativeLQR, Task, zeros
        rec = {}
        for name in ("c3", "c2"):
            system, task, model, sp = make_workload(name, precision="f64", device=R.local_rank, seed=0)
            for noise in ("numpy", "device"):
                np.random.seed(0)
                ctl = MPPI(system, task, model, horizon=sp["horizon"], num_path=sp["num_path"], sigma=1.0,
                           lmda=1.0, noise=noise)
                obs = task.get_init_obs()
                one = zeros(system, 1)
                one.obs[0, :] = obs
                cs = ctl.traj_to_state(one)
                for _ in range(30):
                    u, cs = ctl.run(cs, obs)
                n = 400 if name == "c3" else 1500
                t0 = time.perf_counter()
                for _ in range(n):
                    u, cs = ctl.run(cs, obs)
                dt = time.perf_counter() - t0
                rec["%s_mppi_run_noise_%s" % (name, noise)] = {"calls_per_s": n / dt, "ms_per_call": 1e3 * dt / n,
                                                               "calls": n}
        system, task, model, sp = make_workload("c3", precision="f64", device=R.local_rank, seed=0)
        nu_ = sp["nu"]
        for tag, bnd in (("free", None), ("bounded_0.25", 0.25)):
            t2 = Task(system)
            t2.set_cost(task.get_cost())
            if bnd is not None:
                t2.set_ctrl_bounds(np.full(nu_, -bnd), np.full(nu_, bnd))
            ctl = IterativeLQR(system, t2, model, 50)
            obs = task.get_init_obs()
            cs = np.concatenate([obs, np.zeros(nu_)])
            ctl.run(cs, obs)
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                ctl.run(cs, obs)
            dt = time.perf_counter() - t0
            rec["c4_ilqr_run_%s" % tag] = {"ms_per_call": 1e3 * dt / reps, "iterations": int(ctl.last_iters),
                                           "calls": reps}
        rec["note"] = ("Controller.run(): host arrays in, host control out; noise 'numpy' = the reference's own "
                       "draw (bit-identical stream, generated on the device), the mode parity is graded in")
        return rec

    def f32_vs_f64_parity():
        """One solve of the same problem, same numpy-drawn noise, in both precisions: the max
        relative difference of the per-sample costs and of the updated control sequence."""
        res = {}
        for prec in ("f64", "f32"):
            h, plan, task, spec = build_plan(prec, 1)
            N, H, nu = spec["num_path"], spec["horizon"], spec["nu"]
            r = np.random.default_rng(7)
            plan.upload(spec.get("x0", task.get_init_obs()), r.normal(size=H * nu),
                        r.normal(size=N * H * nu))
            plan.solve()
            a, _, c, _ = plan.download(costs=True)
            res[prec] = (a, c)
            plan.close()
            h.close()
        rel = lambda x, y: float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
        return {"cost_rel_err": rel(res["f32"][1], res["f64"][1]),
                "act_sequence_rel_err": rel(res["f32"][0], res["f64"][0]), "tolerance": 1e-4}

    default_line = world == 1 and args.workload == "c3" and args.precision == "f64" and not args.no_extras
    elapsed, kt, info, spec, n_pre = timed_run(args.precision, args.steps, args.warmup, args.preheat,
                                               windows=max(1, args.repeats) if default_line else 1)
    nx, nu, N, H, B = spec["nx"], spec["nu"], spec["num_path"], spec["horizon"], batch
    sharded = None
    if world > 1 and args.workload == "c3" and not args.no_extras:
        sharded = c5_sharded_record(args, R)             # collective: every rank calls it

    if rank == 0:
        solves = world * args.steps * B
        value = solves / elapsed
        rollout_s = kt["rollout_ms"] * 1e-3
        if "linear" in spec:          # algorithmic work of x' = A x + B u, not of its staging
            info["flops"] = float(B * N * H * 2 * nx * (nx + nu))
        if "sindy" in spec:           # library evaluation + Theta Xi' (VALU path: no MFMA roofline)
            info["flops"] = float(B * N * H * 2 * nx * spec["sindy"]["n_feat"])
        achieved = info["flops"] / rollout_s / 1e12 if rollout_s > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        traffic = measured_traffic(args, B, "mppi_rollout")
        out = {
            "metric": "MPC solves/sec (MPPI, n_samples x horizon rollouts + update per solve)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "preheat_steps": n_pre, "preheat_s": args.preheat,
            "config": {"workload": "%s: %s; random-weight model, QuadCost Q=I R=0.01I F=I, "
                                   "sigma=1 lmda=1, %d independent solve(s) per step per GPU, "
                                   "noise=%s" % (args.workload, spec["label"], B, args.noise),
                       "n_samples": N, "horizon": H, "state_dim": nx, "ctrl_dim": nu,
                       "hidden": spec["hidden"], "parallelism": "independent solves per GPU (dp%d)" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel": "mppi_rollout4_kernel" if info["samples_per_wg"] == 4 else "mppi_rollout_kernel",
                         "kernel_ms": kt["rollout_ms"],
                         "update_kernel_ms": kt["update_ms"], "launches_timed": kt["count"], "event_stride": EVENT_STRIDE,
                         "algorithmic_flops_per_launch": info["flops"],
                         "algorithmic_bytes_per_launch": info["bytes"],
                         "workgroups": info["workgroups"], "samples_per_workgroup": info["samples_per_wg"]},
        }
        if len(spec["window_rates"]) > 1:
            wr = np.array(spec["window_rates"])
            out["repeat_windows"] = {"windows": len(wr), "steps_each": args.steps, "unit": "solves/s",
                                     "median": float(np.median(wr)), "min": float(wr.min()), "max": float(wr.max()),
                                     "note": "value = the first window; the same window repeated back to back"}
        if sharded is not None:
            out["c5_sharded"] = sharded
        if world == 1 and args.precision == "f64" and not args.no_extras:
            # the exact-f32 MFMA mode of the same kernel (v_mfma_f32_16x16x4_f32): reported next to
            # the f64 headline, with its measured deviation from the f64 solve on identical inputs
            s32, w32 = max(1, args.steps // 2), max(1, args.warmup // 2)
            e32, k32, i32, _, _ = timed_run("f32", s32, w32, min(args.preheat, 0.3))
            a32 = i32["flops"] / (k32["rollout_ms"] * 1e-3) / 1e12
            out["f32_fast_mode"] = {"value": s32 * B / e32, "unit": "solves/s",
                                    "kernel_ms": k32["rollout_ms"], "achieved_tflops": a32,
                                    "frac_of_f32_mfma_peak": a32 / PEAK_TFLOPS["f32"],
                                    "vs_f64_solve": f32_vs_f64_parity()}
            # eight independent solves per launch -- what a tuner's batch looks like: enough tiles for two 16-row
            # f32 workgroups per CU (one's serial chain behind the other's MFMAs) / one 32-row f64 workgroup per CU
            for prec, key in (("f32", "f32_fast_mode"), ("f64", "batch8_f64")):
                sb = max(1, args.steps // 8)
                e8, k8, i8, _, _ = timed_run(prec, sb, max(1, args.warmup // 4), 0.2, nb=8)
                a8 = i8["flops"] / (k8["rollout_ms"] * 1e-3) / 1e12
                rec8 = {"workload": "8 independent config-3 solves per launch (%d-row tiles)" % i8["samples_per_wg"],
                        "value": world * sb * 8 / e8, "unit": "solves/s", "kernel_ms": k8["rollout_ms"],
                        "achieved_tflops": a8, "frac_of_mfma_peak": a8 / PEAK_TFLOPS[prec]}
                if prec == "f32":
                    out["f32_fast_mode"]["batch8"] = rec8
                else:
                    out[key] = rec8
        if default_line:
            # the other BASELINE configurations and the user-visible call rates, measured inside the
            # same driver-timed process; full records: --workload c2 / c4 / c5
            sub = argparse.Namespace(**vars(args))
            sub.no_cpu_baseline, sub.no_extras, sub.batch = True, False, 0
            recs = {"c2": mppi_record("c2", 400, 40)}
            sub.trimmed_cpu_seconds = 0.0 if args.no_cpu_baseline else 4.0
            for wl, st in (("c4", 5), ("c5", 2)):
                sub.workload, sub.steps, sub.warmup, sub.preheat = wl, st, 1, 0.0
                r = secondary_workload(sub, R, emit=False)
                recs[wl] = {k: v for k, v in r.items()
                            if k not in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "preheat_steps")}
            recs["ilqr_eval"] = ilqr_eval_record(args, R)
            recs["model_axis"] = model_axis_record(args, R)
            recs["dropin"] = dropin_record()
            out["sub_records"] = recs
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_mppi(args.workload, spec, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    R.close()


if __name__ == "__main__":
    main()
