#!/usr/bin/env python3
"""Headline benchmark: MPC solves/sec of the MPPI inner solve on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2] [--precision f64|f32]

One "step" = one complete MPPI solve of the named BASELINE.json configuration (default c3:
HalfCheetah-shaped MLP 2x256, 4096 samples x 30 horizon): fresh device-resident noise, batched
surrogate rollout with stage/action/terminal costs, softmin weighting and warm-start update --
i.e. everything MPPI.run() does between receiving an observation and returning a control
(reference: autompc/control/mppi.py:120-168).  Inputs are resident in HBM when the timed region
starts.  For N > 1 every rank (one process per GPU, launched by torch.distributed.run) solves
its own independent problems -- the tuning use case shards candidate controllers, a single
solve does not shard -- so scaling is weak and there is no collective on
the data path; ranks only meet at the barriers that bracket the timed region and at the final
max-over-ranks reduction of the elapsed time.

Prints ONE JSON line on rank 0.
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
    ap.add_argument("--batch", type=int, default=1, help="independent solves per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the f32 fast-mode side report")
    ap.add_argument("--cpu-solves", type=int, default=0, help="0 = auto (about 10-30 s)")
    return ap.parse_args()


def cpu_baseline(workload, spec, n_solves):
    """The oracle (numpy restatement of the reference, strict_reference=True: per-step
    pred_batch + the per-particle Python cost loop of mppi.py:73-78) timed on the host."""
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
    u, cs = ctl.run(cs, x0)                      # warm-up (BLAS threads, caches)
    if n_solves <= 0:
        t0 = time.perf_counter()
        u, cs = ctl.run(cs, x0)
        one = time.perf_counter() - t0
        n_solves = int(max(2, min(50, round(12.0 / max(one, 1e-3)))))
    t0 = time.perf_counter()
    for _ in range(n_solves):
        u, cs = ctl.run(cs, x0)
    dt = time.perf_counter() - t0
    return {"value": n_solves / dt, "unit": "solves/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d consecutive %s solves (oracle, numpy f64 + per-particle Python cost loop, "
                      "OpenBLAS threads = host cores), %.1f s" % (n_solves, workload, dt)}


def measured_traffic(args, batch):
    """HBM bytes per rollout launch from the committed PMC passes (profiles/r01_hbm_traffic.json,
    written by tools/summarize_profiles.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    runs of this same command).  Counters cannot be collected from inside the timed process, so
    the figure is reported only for the configuration it was measured on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic.json")
    if not (args.workload == "c3" and args.precision == "f64" and batch == 1 and args.noise == "device"
            and os.path.exists(path)):
        return None
    try:
        with open(path) as f:
            d = json.load(f)
        for name, v in d["kernels"].items():
            if "mppi_rollout_kernel<double" in name:
                return {"bytes": v["bytes"], "source": "profiles/r01_hbm_traffic.json (%s)" % d["method"]}
    except (OSError, ValueError, KeyError):
        pass
    return None


def secondary_workload(args, rank, local_rank, world):
    """c4 / c5: the other BASELINE configurations, same timing contract (barrier + sync on both
    sides, max over ranks), reported with their own unit of work."""
    import torch
    import torch.distributed as dist
    from autompc_amd import _lib
    from autompc_amd.synthetic import make_workload
    system, task, model, spec = make_workload("c3", precision=args.precision, device=local_rank)
    nx, nu = spec["nx"], spec["nu"]
    B = args.batch if args.batch > 1 else (64 if args.workload == "c5" else 256)
    extra = {}

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.workload == "c4":
        h = _lib.Handle(local_rank, args.precision)
        model.stage_into(h)
        Q, R, F = task.get_cost().get_cost_matrices()
        h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
        plan = _lib.IlqrPlan(h, B, 50, system.dt)
        rng = np.random.default_rng(rank)
        x0 = rng.uniform(-0.1, 0.1, size=(B, nx))
        ug = np.zeros((B, 50, nu))
        iters = []

        def step(i):
            out = plan.solve(x0, ug, max_iter=50)
            iters.append(float(out["iters"].mean()))
        label = "c4: HalfCheetah MLP 2x256, iLQR horizon 50, %d independent problems per step per GPU" % B
        unit_per_step = B
        metric, unit = "MPC solves/sec (iLQR, full compute_ilqr_default per solve)", "solves/s"
    else:
        from autompc_amd.tuning import CandidateEvaluator, evaluate_sharded, random_candidates
        task.set_num_steps(200)
        # BASELINE config 5: one candidate list for the whole job, contiguous shards of B per GPU,
        # scores exchanged with one all-gather (RCCL over xGMI under the nccl backend)
        cands = random_candidates(system, B * world, seed=0)
        ev = CandidateEvaluator(system, task, model, precision=args.precision, device=local_rank)

        def step(i):
            scores = evaluate_sharded(
                lambda shard, lo: ev.evaluate(shard, n_steps=200, seed=i, index_offset=lo), cands)
            if not np.all(np.isfinite(scores)) or scores.shape[0] != B * world:
                raise RuntimeError("candidate scores incomplete")
        label = ("c5: %d tuning candidates (MPPI horizon/sigma/lmda/num_path + QuadCost weights from "
                 "the reference's config ranges) x 200-step closed loop per step per GPU" % B)
        unit_per_step = B * 200
        metric, unit = "MPC solves/sec (MPPI inside the batched closed-loop candidate evaluator)", "solves/s"
    steps, warm = max(1, min(args.steps, 10)), max(1, min(args.warmup, 2))
    for i in range(warm):
        step(i)
    iters.clear() if args.workload == "c4" else None
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warm + i)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        mlp_macs = sum(a * b for a, b in zip([nx + nu] + spec["hidden"], spec["hidden"] + [nx]))
        if args.workload == "c4":
            # algorithmic work of one iteration of one problem: Jacobian chain over H
            # rows + the 10-candidate line-search rollout + the accepted-trajectory forward pass
            it = float(np.mean(iters))
            hid = sum(a * b for a, b in zip(spec["hidden"], spec["hidden"][1:]))
            jac = 50 * 2 * nx * (hid + spec["hidden"][0] * (nx + nu))
            per_iter = jac + 10 * 50 * 2 * mlp_macs + 50 * 2 * mlp_macs
            extra["mean_iterations_per_solve"] = it
            extra["algorithmic_tflops"] = world * steps * B * it * per_iter / elapsed / 1e12
        else:
            # every MPPI solve of candidate c is N_c x H_c model steps (+ the stage costs)
            per_ctrl_step = sum(c["num_path"] * c["horizon"] for c in cands) / world
            flops = per_ctrl_step * (2 * mlp_macs + 2 * (nx * nx + nx) + 2 * nu * nu + 2 * nu)
            extra["algorithmic_tflops"] = world * steps * 200 * flops / elapsed / 1e12
        extra["mfma_peak_tflops"] = PEAK_TFLOPS[args.precision]
        out = {"metric": metric, "value": world * steps * unit_per_step / elapsed, "unit": unit,
               "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": 1e3 * elapsed / steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.precision, "data": "synthetic",
               "config": {"workload": label, "parallelism": "independent problems per GPU (dp%d)" % world},
               "roofline": None, **extra}
        if not args.no_cpu_baseline and world == 1 and args.workload == "c4":
            from oracle.costs import QuadCostOracle
            from oracle.ilqr import ILQROracle
            from oracle.mlp import MLPOracle, make_params
            p = spec["params"]
            om = MLPOracle(system, make_params(p["weights"], p["biases"], "relu", p["xu_means"],
                                               p["xu_std"], p["dy_means"], p["dy_std"]))
            orc = ILQROracle(om, QuadCostOracle(np.eye(nx), 0.01 * np.eye(nu), np.eye(nx), np.zeros(nx)),
                             system.dt, 50)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 10.0:
                orc.solve(x0[n % B], np.zeros((50, nu)))
                n += 1
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": os.cpu_count(),
                                   "kind": "port", "sample": "%d iLQR solves (oracle, numpy f64), %.1f s" % (n, dt)}
        print(json.dumps(out))


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


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    # Plumbing test hooks (tools/gpu_round.sh): run a 2-rank job on a 1-GPU box by mapping every
    # rank to one device and using gloo for the barriers.  Never set by the driver.
    if "AMPC_BENCH_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["AMPC_BENCH_FORCE_DEVICE"])
    backend = os.environ.get("AMPC_BENCH_BACKEND", "nccl")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d device(s) are visible "
                         "(--gpus %d)" % (rank, local_rank, torch.cuda.device_count(), args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.workload in ("c4", "c5"):
        secondary_workload(args, rank, local_rank, world)
        if world > 1:
            dist.destroy_process_group()
        return

    from autompc_amd import _lib
    from autompc_amd.synthetic import make_workload

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def build_plan(precision, batch):
        system, task, model, spec = make_workload(args.workload, precision=precision,
                                                  device=local_rank, seed=0)
        stream = torch.cuda.current_stream().cuda_stream
        h = _lib.Handle(local_rank, precision, stream=stream)
        model.stage_into(h)
        Q, R, F = task.get_cost().get_cost_matrices()
        h.set_quad_costs(Q, R, F, task.get_cost().get_goal())
        bounds = task.get_ctrl_bounds()
        h.set_ctrl_bounds(bounds[:, 0], bounds[:, 1])
        N, H = spec["num_path"], spec["horizon"]
        plan = _lib.MppiPlan(h, [N] * batch, [H] * batch, [1.0] * batch, [1.0] * batch)
        return h, plan, task, spec

    def timed_run(precision, steps, warmup):
        """W untimed + K timed solves, bracketed by barrier + synchronize; max over ranks."""
        h, plan, task, spec = build_plan(precision, args.batch)
        nx, nu, N, H = spec["nx"], spec["nu"], spec["num_path"], spec["horizon"]
        B = args.batch
        rng = np.random.default_rng(1000 + rank)
        x0 = np.tile(spec.get("x0", task.get_init_obs()), (B, 1))
        if "linear" not in spec:      # (an ARX state repeats the observation: leave it consistent)
            x0 = x0 + rng.uniform(-0.01, 0.01, size=(B, nx))
        np.random.seed(rank)
        act0 = np.random.normal(size=(B * H * nu))
        eps0 = np.random.normal(size=(B * N * H * nu)) if args.noise == "resident" else None
        plan.upload(x0, act0, eps0)
        plan.set_outputs(keep_eps_out=False)   # nothing downloads the clipped noise here
        info = plan.info()

        def step(i):
            if args.noise == "device":
                plan.generate_eps(rank, i)
            plan.solve()
        for i in range(warmup):
            step(i)
        plan.set_timing(True)
        sync_all()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        sync_all()
        elapsed = time.perf_counter() - t0
        kt = plan.timing()
        plan.set_timing(False)
        _, u, _, _ = plan.download(act_seq=False, u=True)
        if not np.all(np.isfinite(u)):
            raise RuntimeError("non-finite control returned by the solve")
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64,
                             device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        plan.close()
        h.close()
        return elapsed, kt, info, spec

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

    elapsed, kt, info, spec = timed_run(args.precision, args.steps, args.warmup)
    nx, nu, N, H, B = spec["nx"], spec["nu"], spec["num_path"], spec["horizon"], args.batch

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
        traffic = measured_traffic(args, B)
        out = {
            "metric": "MPC solves/sec (MPPI, n_samples x horizon rollouts + update per solve)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%s: %s; random-weight model, QuadCost Q=I R=0.01I F=I, "
                                   "sigma=1 lmda=1, %d independent solve(s) per step per GPU, "
                                   "noise=%s" % (args.workload, spec["label"], B, args.noise),
                       "n_samples": N, "horizon": H, "state_dim": nx, "ctrl_dim": nu,
                       "hidden": spec["hidden"], "parallelism": "independent solves per GPU (dp%d)" % world},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel": "mppi_rollout_kernel", "kernel_ms": kt["rollout_ms"],
                         "update_kernel_ms": kt["update_ms"], "launches_timed": kt["count"],
                         "algorithmic_flops_per_launch": info["flops"],
                         "algorithmic_bytes_per_launch": info["bytes"],
                         "workgroups": info["workgroups"], "samples_per_workgroup": info["samples_per_wg"]},
        }
        if world == 1 and args.precision == "f64" and not args.no_extras:
            # the exact-f32 MFMA mode of the same kernel (v_mfma_f32_16x16x4_f32): reported next to
            # the f64 headline, with its measured deviation from the f64 solve on identical inputs
            e32, k32, i32, _ = timed_run("f32", max(1, args.steps // 2), max(1, args.warmup // 2))
            a32 = i32["flops"] / (k32["rollout_ms"] * 1e-3) / 1e12
            out["f32_fast_mode"] = {"value": max(1, args.steps // 2) * B / e32, "unit": "solves/s",
                                    "kernel_ms": k32["rollout_ms"], "achieved_tflops": a32,
                                    "frac_of_f32_mfma_peak": a32 / PEAK_TFLOPS["f32"],
                                    "vs_f64_solve": f32_vs_f64_parity()}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload, spec, args.cpu_solves)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
