"""Randomised parity sweep on the GPU box: many random model shapes / solver settings, HIP path
(through the Python mirrors -> C ABI) against the oracle.  Prints the worst relative errors and
exits non-zero on a violation.  Usage: python tools/fuzz_gpu.py [n_cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from autompc_amd import MLP, MPPI, IterativeLQR, QuadCost, System, Task
from oracle import mlp as omlp
from oracle.costs import QuadCostOracle, SumCostOracle
from oracle.ilqr import ILQROracle
from oracle.mlp import MLPOracle
from oracle.mppi import MPPIOracle


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = {"pred": 0.0, "jac": 0.0, "mppi_cost": 0.0, "mppi_act": 0.0, "ilqr": 0.0}
    bad = []
    n_jit = [0]
    for case in range(n_cases):
        nx = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 17, 18, 19, 20, 21, 25, 32, 33, 40, 50, 64]))
        nu = int(rng.choice([1, 2, 3, 6, 9, 16]))
        if nx <= 32 and nx + nu > 48:
            nu = 48 - nx
        if nx > 32 and nx + nu > 80:          # (the WIDE tile: first-layer K up to 80)
            nu = 80 - nx
        nl = int(rng.integers(1, 5))
        hidden = [int(rng.choice([16, 33, 64, 100, 128, 150, 192, 256])) for _ in range(nl)]
        act = str(rng.choice(["relu", "tanh", "sigmoid", "selu"]))
        prec = "f64" if rng.random() < 0.7 else "f32"
        tol = 1e-9 if prec == "f64" else 2e-4
        os.environ["AMPC_MT"] = str(rng.choice([0, 1, 2, 4]))
        # a quarter of the cases wait for the kernels compiled at run time for the case's shape
        # (csrc/jit_host.hpp) and run on those; the rest stay on the run-time-shape kernels
        use_jit = rng.random() < 0.25
        os.environ["AMPC_JIT"] = "1" if use_jit else "0"
        # the line search of the case's iLQR solves: four-row passes side by side (few problems), in sequence,
        # all step sizes in one twelve-row pass (ilqr_lsw.hpp), or picked per poll -- one oracle for all
        ls_par, ls_rb = str(rng.choice([0, 1])), str(rng.choice([0, 1, 3]))
        os.environ["AMPC_LS4_PAR"], os.environ["AMPC_LS4_RB"] = ls_par, ls_rb
        system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)
        p = omlp.random_params(nx, nu, hidden, act, seed=int(rng.integers(1 << 30)))
        p["xu_means"] = rng.normal(scale=0.2, size=nx + nu)
        p["xu_std"] = rng.uniform(0.5, 2.0, size=nx + nu)
        p["dy_means"] = rng.normal(scale=0.02, size=nx)
        p["dy_std"] = rng.uniform(0.05, 0.2, size=nx)
        m = MLP(system, n_hidden_layers=nl, nonlintype=act, precision=prec,
                **{"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)})
        m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
        m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
        tag = "case %d nx=%d nu=%d hidden=%s %s %s MT=%s jit=%d ls=%s%s" % (case, nx, nu, hidden, act, prec, os.environ["AMPC_MT"], use_jit,
                                                                             ls_par, ls_rb)
        try:
            n = int(rng.choice([1, 7, 16, 33, 200]))
            s, c = rng.normal(size=(n, nx)), rng.normal(size=(n, nu))
            e = rel(m.pred_batch(s, c), omlp.pred_batch(p, s, c))
            worst["pred"] = max(worst["pred"], e / tol)
            o, jx, ju = m.pred_diff_batch(s, c)
            eo, ejx, eju = omlp.pred_diff_batch(p, s, c)
            ej = max(rel(jx, ejx), rel(ju, eju))
            worst["jac"] = max(worst["jac"], ej / (10 * tol))
            if e > tol or ej > 10 * tol:
                bad.append((tag, "pred/jac", e, ej))
            # MPPI
            Q = np.diag(rng.uniform(0.5, 2.0, size=nx)) if rng.random() < 0.6 else \
                (lambda A: A @ A.T / nx + 0.1 * np.eye(nx))(rng.normal(size=(nx, nx)))
            R = np.diag(rng.uniform(0.01, 0.1, size=nu))
            F = np.diag(rng.uniform(0.5, 2.0, size=nx))
            goal = rng.normal(scale=0.1, size=nx)
            # a third of the cases: a SUM of quadratic terms with different goals (sum_cost.py:49-54) -- one
            # affine-quadratic block on the device, the term-by-term fan-out in the oracle
            hip_cost, orc_cost = QuadCost(system, Q, R, F, goal=goal), QuadCostOracle(Q, R, F, goal)
            if rng.random() < 0.35:
                terms = [(Q, R, F, goal)]
                for _ in range(int(rng.integers(1, 3))):
                    Wq = rng.normal(size=(nx, nx))
                    terms.append((Wq @ Wq.T / nx * float(rng.uniform(0.05, 0.5)) if rng.random() < 0.5
                                  else np.diag(rng.uniform(0.05, 0.5, size=nx)), np.diag(rng.uniform(0.0, 0.05, size=nu)),
                                  np.diag(rng.uniform(0.0, 0.5, size=nx)), rng.normal(scale=0.3, size=nx)))
                hip_cost = QuadCost(system, *terms[0][:3], goal=terms[0][3])
                for t_ in terms[1:]:
                    hip_cost = hip_cost + QuadCost(system, *t_[:3], goal=t_[3])
                orc_cost = SumCostOracle.from_arrays(*zip(*terms))
                tag += " sumcost%d" % len(terms)
            # a quarter of the cases: threshold / box terms on top (MPPI only -- they have no gradient; round 5,
            # ampc_set_indicator_costs): integer steps in the cost that tell samples apart
            mppi_hip, mppi_orc = hip_cost, orc_cost
            if rng.random() < 0.25:
                from autompc_amd import BoxThresholdCost, ThresholdCost
                from oracle.costs import BoxCostOracle, ThresholdCostOracle
                ind_h, ind_o = [], []
                for _ in range(int(rng.integers(1, 4))):
                    if rng.random() < 0.5:
                        a = int(rng.integers(0, nx)); b = int(rng.integers(a + 1, nx + 1))
                        g2, thr = goal + rng.normal(scale=0.02, size=nx), float(rng.uniform(0.02, 0.3))
                        ind_h.append(ThresholdCost(system, g2, [a, b], thr)); ind_o.append(ThresholdCostOracle(g2, [a, b], thr))
                    else:
                        lim = np.stack([goal - rng.uniform(0.02, 0.4, size=nx), goal + rng.uniform(0.02, 0.4, size=nx)], axis=1)
                        lim[rng.random(nx) < 0.3, 0] = -np.inf
                        ind_h.append(BoxThresholdCost(system, lim)); ind_o.append(BoxCostOracle(lim))
                for t_ in ind_h:
                    mppi_hip = mppi_hip + t_
                mppi_orc = SumCostOracle((orc_cost.terms if isinstance(orc_cost, SumCostOracle) else [orc_cost]) + ind_o)
                tag += " ind%d" % len(ind_h)
                worst.setdefault("mppi_indicator_cases", 0.0)
                worst["mppi_indicator_cases"] += 1
            task = Task(system)
            task.set_cost(mppi_hip)
            lo, hi = -float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.3, 1.5))
            task.set_ctrl_bounds(np.full(nu, lo), np.full(nu, hi))
            N, H = int(rng.choice([17, 64, 100, 300])), int(rng.integers(2, 20))   # (H = 1 raises in the reference: a[-2])
            sigma, lmda = float(rng.uniform(0.2, 1.5)), float(rng.uniform(0.2, 2.0))
            seed = int(rng.integers(1 << 30))
            omodel = MLPOracle(system, p)
            np.random.seed(seed)
            orc = MPPIOracle(omodel, mppi_orc, np.tile([lo, hi], (nu, 1)),
                             horizon=H, num_path=N, sigma=sigma, lmda=lmda)
            np.random.seed(seed)
            ctl = MPPI(system, task, m, horizon=H, num_path=N, sigma=sigma, lmda=lmda)
            if use_jit:
                ctl._device()
                ctl._handle.jit_wait()
                n_jit[0] += ctl._device().kernel_kind() == 2
            obs = rng.uniform(-0.1, 0.1, size=nx)
            cs = np.concatenate([obs, np.zeros(nu)])
            st = np.random.get_state()
            uo, _ = orc.run(cs, obs)
            np.random.set_state(st)
            uh, _ = ctl.run(cs, obs, return_details=True)
            ec, ea = rel(ctl.last_costs, orc.last_costs), rel(ctl.act_sequence, orc.act_sequence)
            worst["mppi_cost"] = max(worst["mppi_cost"], ec / tol)
            worst["mppi_act"] = max(worst["mppi_act"], ea / (100 * tol))
            if ec > tol or ea > 100 * tol:
                bad.append((tag + " N=%d H=%d" % (N, H), "mppi", ec, ea))
                # keep the case (tools/fuzz_replay.py) and say what differs: the noise, or single samples' costs
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez("gpurun_out/fuzz_case_%d.npz" % case, nx=nx, nu=nu, hidden=hidden, act=act, prec=prec,
                         mt=os.environ["AMPC_MT"], jit=int(use_jit), Q=Q, R=R, F=F, goal=goal, lo=lo, hi=hi, N=N, H=H,
                         sigma=sigma, lmda=lmda, seed=seed, obs=obs, n_layers=nl,
                         **{"W%d" % i: w for i, w in enumerate(p["weights"])},
                         **{"b%d" % i: b for i, b in enumerate(p["biases"])},
                         xu_means=p["xu_means"], xu_std=p["xu_std"], dy_means=p["dy_means"], dy_std=p["dy_std"],
                         costs_ref=orc.last_costs, eps_ref=orc.last_eps)
                de = np.abs(ctl.last_eps - orc.last_eps)
                dc = np.abs(ctl.last_costs - orc.last_costs) / np.abs(orc.last_costs)
                print("  mppi detail: kernel kind %d, rows/wg %d, max |d eps| %.3e at %s, samples with cost error > tol: %s"
                      % (ctl._device().kernel_kind(), ctl._device().info()["samples_per_wg"], de.max(),
                         np.unravel_index(np.argmax(de), de.shape), np.nonzero(dc > tol)[0][:20].tolist()), flush=True)
            # iLQR (f64 only; a few iterations, compare the first accepted trajectory loosely)
            if prec == "f64" and case % 3 == 0 and nx + nu <= 45:
                Hh = int(rng.integers(3, 15))
                t2 = Task(system)
                t2.set_cost(hip_cost)
                il = IterativeLQR(system, t2, m, Hh)
                if use_jit:
                    il._device()
                    il._handle.jit_wait()
                n_it = int(rng.integers(1, 7))
                oil = ILQROracle(omodel, orc_cost, system.dt, Hh, max_iter=n_it)
                r1 = il._device().solve(obs[None, :], np.zeros((1, Hh, nu)), n_it)
                co, so, uo2, Ko, ko = oil.solve(obs, np.zeros((Hh, nu)))
                ei = max(rel(r1["states"][0], so), rel(r1["Ks"][0], Ko))
                worst["ilqr"] = max(worst["ilqr"], ei / 1e-6)
                if ei > 1e-6:
                    bad.append((tag + " H=%d" % Hh, "ilqr", ei, 0.0))
                # the same solve as problem 2 of 5 streamed through a two-slot queue: bit-identical
                if case % 2 == 0:
                    from autompc_amd import _lib
                    xs = np.stack([obs + 0.01 * k for k in (-2, -1, 0, 1, 2)])
                    qp = _lib.IlqrPlan(il._handle, 2, Hh, system.dt, clip_to_bounds=False,
                                       terminal_goal=il._terminal_goal)
                    rq = qp.solve_queue(xs, max_iter=n_it)
                    qp.close()
                    same = all(np.array_equal(rq[k][2], r1[k][0]) for k in ("states", "ctrls", "Ks", "ks", "iters"))
                    worst["ilqr_queue"] = max(worst.get("ilqr_queue", 0.0), 0.0 if same else 1e9)
                    if not same:
                        bad.append((tag + " H=%d" % Hh, "ilqr queue differs from the one-problem solve", 0.0, 0.0))
                    # per-problem horizons through ONE plan (round 5, ampc_ilqr_solve_queue_var): each problem bit for
                    # bit what a one-problem plan of its own horizon gives
                    hz = rng.integers(2, Hh + 1, size=5).astype(np.int32)
                    qp = _lib.IlqrPlan(il._handle, 2, Hh, system.dt, clip_to_bounds=False, terminal_goal=il._terminal_goal)
                    rv = qp.solve_queue(xs, max_iter=n_it, horizon=hz)
                    qp.close()
                    same = True
                    for j in (0, 3):
                        Hj = int(hz[j])
                        one = _lib.IlqrPlan(il._handle, 1, Hj, system.dt, clip_to_bounds=False, terminal_goal=il._terminal_goal)
                        rj = one.solve(xs[j], np.zeros((Hj, nu)), n_it)
                        one.close()
                        same = same and np.array_equal(rv["states"][j, :Hj + 1], rj["states"][0]) and \
                            all(np.array_equal(rv[k][j, :Hj], rj[k][0]) for k in ("ctrls", "Ks", "ks")) and \
                            np.array_equal(rv["iters"][j], rj["iters"][0]) and not rv["ctrls"][j, Hj:].any()
                    worst["ilqr_var_horizon"] = max(worst.get("ilqr_var_horizon", 0.0), 0.0 if same else 1e9)
                    if not same:
                        bad.append((tag + " H=%d hz=%s" % (Hh, hz.tolist()), "per-problem horizons differ from one-horizon plans", 0.0, 0.0))
        except Exception as ex:          # noqa: BLE001 -- report and continue
            bad.append((tag, "exception", repr(ex)[:200], 0.0))
    # ---- linear models, closed loop and device scoring ------------------------------------------
    from autompc_amd import Koopman, simulate
    from autompc_amd.costs import BoxThresholdCost, ThresholdCost, cost_terms
    from autompc_amd.tuning import CandidateEvaluator
    from oracle.costs import score_terms
    from oracle.linear import LinearOracle
    worst.update({"lin_pred": 0.0, "lin_mppi": 0.0, "loop_score": 0.0})
    os.environ["AMPC_MT"] = "0"
    os.environ.pop("AMPC_LS4_PAR", None)
    os.environ.pop("AMPC_LS4_RB", None)
    for case in range(max(4, n_cases // 10)):
        ns, nu = int(rng.integers(1, 65)), int(rng.integers(1, 9))     # 33..64: the four-output-tile path
        if case % 3 == 2:
            ns = int(rng.integers(65, 257))                              # 65..256: csrc/linear_kernels.hpp
        if ns <= 32 and ns + nu > 48:
            nu = 48 - ns
        no = int(rng.integers(1, min(ns, 64) + 1))
        system = System(["x%d" % i for i in range(no)], ["u%d" % i for i in range(nu)], dt=0.05)
        S = rng.normal(size=(ns, ns))
        A = np.eye(ns) * 0.9 + 0.1 * (S - S.T) / max(1.0, np.sqrt(ns))
        B = rng.normal(scale=0.3, size=(ns, nu))
        tag = "linear case %d ns=%d no=%d nu=%d" % (case, ns, no, nu)
        try:
            m = Koopman(system)                      # used as a carrier of (A, B) on the device
            m.set_parameters({"A": A, "B": B})
            m._apply_basis = lambda o, ns=ns: np.concatenate([np.asarray(o), np.zeros(ns - len(o))])
            type(m).state_dim = property(lambda self: self.A.shape[0])
            orc_m = LinearOracle(system, A, B)
            orc_m.state_dim = ns
            orc_m.update_state = lambda st, c, o, ns=ns: np.concatenate([np.asarray(o), np.zeros(ns - len(o))])
            s_, c_ = rng.normal(size=(9, ns)), rng.normal(size=(9, nu))
            e = rel(m.pred_batch(s_, c_), orc_m.pred_batch(s_, c_))
            o_, jx, ju = m.pred_diff_batch(s_, c_)
            e = max(e, rel(jx[3], A), rel(ju[5], B))
            worst["lin_pred"] = max(worst["lin_pred"], e / 1e-12)
            if e > 1e-12:
                bad.append((tag, "linear pred/jac", e, 0.0))
            Q, R, F = np.diag(rng.uniform(0.5, 2, size=no)), np.diag(rng.uniform(0.01, 0.1, size=nu)), np.eye(no)
            goal = rng.normal(scale=0.1, size=no)
            task = Task(system)
            task.set_cost(QuadCost(system, Q, R, F, goal=goal))
            task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
            N, H = int(rng.choice([32, 100])), int(rng.integers(2, 12))
            seed = int(rng.integers(1 << 30))
            np.random.seed(seed)
            orc = MPPIOracle(orc_m, QuadCostOracle(Q, R, F, goal), np.tile([-1.0, 1.0], (nu, 1)),
                             horizon=H, num_path=N, sigma=0.5, lmda=0.8)
            np.random.seed(seed)
            ctl = MPPI(system, task, m, horizon=H, num_path=N, sigma=0.5, lmda=0.8)
            obs = rng.uniform(-0.5, 0.5, size=no)
            cs = np.concatenate([obs, np.zeros(ns - no), np.zeros(nu)])
            st = np.random.get_state()
            uo, _ = orc.run(cs, obs)
            np.random.set_state(st)
            uh, _ = ctl.run(cs, obs, return_details=True)
            e = max(rel(ctl.last_costs, orc.last_costs), rel(uh, uo) / 100)
            worst["lin_mppi"] = max(worst["lin_mppi"], e / 1e-9)
            if e > 1e-9:
                bad.append((tag, "linear mppi", e, 0.0))
        except Exception as ex:      # noqa: BLE001
            bad.append((tag, "exception", repr(ex)[:200], 0.0))
    # closed loop with a summed indicator + quadratic task cost, scored on the device
    for case in range(max(3, n_cases // 20)):
        nx, nu = int(rng.integers(2, 9)), int(rng.integers(1, 4))
        system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)
        p = omlp.random_params(nx, nu, [64, 64], "tanh", seed=int(rng.integers(1 << 30)))
        m = MLP(system, n_hidden_layers=2, nonlintype="tanh", hidden_size_1=64, hidden_size_2=64)
        m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
        m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
        goal = rng.normal(scale=0.1, size=nx)
        lim = np.stack([-rng.uniform(0.2, 1.0, size=nx), rng.uniform(0.2, 1.0, size=nx)], axis=1)
        cost = (ThresholdCost(system, goal, [0, int(rng.integers(1, nx + 1))], float(rng.uniform(0.05, 0.5)))
                + BoxThresholdCost(system, lim) + QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), np.eye(nx), goal=goal))
        task = Task(system)
        task.set_cost(cost)
        task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
        task.set_init_obs(rng.uniform(-0.5, 0.5, size=nx))
        task.set_num_steps(int(rng.integers(5, 30)))
        ev = CandidateEvaluator(system, task, m)
        cands = [dict(horizon=int(rng.integers(3, 12)), sigma=0.4, lmda=0.6, num_path=int(rng.choice([48, 100])),
                      Q=np.ones(nx), R=0.1 * np.ones(nu), F=np.ones(nx)) for _ in range(int(rng.integers(1, 6)))]
        try:
            scores, obs_t, ctl_t = ev.evaluate(cands, seed=case, return_trajectories=True)
            terms = cost_terms(cost, nx, nu)
            for b in range(len(cands)):
                ref = score_terms(terms[0], terms[1], obs_t[b], ctl_t[b])
                e = abs(scores[b] - ref) / max(1.0, abs(ref))
                worst["loop_score"] = max(worst["loop_score"], e / 1e-10)
                if e > 1e-10:
                    bad.append(("closed loop case %d cand %d" % (case, b), "score", e, 0.0))
        except Exception as ex:      # noqa: BLE001
            bad.append(("closed loop case %d" % case, "exception", repr(ex)[:200], 0.0))
    # ---- several controller models of one (random, run-time compiled) shape in one candidate batch (round 5) ----
    # a candidate's score is bit for bit what it gets alone with only its own model (MPPI and iLQR evaluators)
    from autompc_amd.tuning import IlqrCandidateEvaluator, random_candidates, random_ilqr_candidates
    os.environ["AMPC_JIT"] = "1"
    worst["model_tables"] = 0.0
    for case in range(max(2, n_cases // 100)):
        nx, nu = int(rng.integers(2, 20)), int(rng.integers(1, 5))
        nl = int(rng.integers(1, 4))
        hidden = [int(rng.choice([32, 64, 100, 128, 192, 256])) for _ in range(nl)]
        act = str(rng.choice(["relu", "tanh"]))
        system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)
        tag = "model table case %d nx=%d nu=%d hidden=%s %s" % (case, nx, nu, hidden, act)
        try:
            models = []
            for k in range(3):
                p = omlp.random_params(nx, nu, hidden, act, seed=int(rng.integers(1 << 30)))
                m = MLP(system, n_hidden_layers=nl, nonlintype=act, **{"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)})
                m.weights, m.biases = [w.copy() for w in p["weights"]], [b.copy() for b in p["biases"]]
                m.xu_means, m.xu_std, m.dy_means, m.dy_std = p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"]
                models.append(m)
            task = Task(system)
            task.set_cost(QuadCost(system, np.eye(nx), 0.01 * np.eye(nu), np.eye(nx)))
            task.set_ctrl_bounds(-np.ones(nu), np.ones(nu))
            task.set_init_obs(rng.uniform(-0.1, 0.1, size=nx))
            task.set_num_steps(int(rng.integers(4, 9)))
            cands = random_candidates(system, 7, seed=case)
            for k, c in enumerate(cands):
                c["model"] = models[k % 3]
                c["num_path"] = int(rng.choice([32, 64, 100]))
            full = CandidateEvaluator(system, task, models[0]).evaluate(cands, seed=4)
            ok = np.all(np.isfinite(full))
            for k in (1, 5):
                own = CandidateEvaluator(system, task, cands[k]["model"], surrogate=models[0])
                ok = ok and own.evaluate([{kk: v for kk, v in cands[k].items() if kk != "model"}], seed=4, index_offset=k)[0] == full[k]
            ic = random_ilqr_candidates(system, 5, seed=case)
            for k, c in enumerate(ic):
                c["Q"], c["R"], c["F"] = c["Q"] ** 0.25, c["R"] ** 0.25, c["F"] ** 0.25
                c["model"] = models[(k + 1) % 3]
            ifull = IlqrCandidateEvaluator(system, task, models[0], max_slots=2).evaluate(ic)
            own = IlqrCandidateEvaluator(system, task, ic[3]["model"], surrogate=models[0])
            ok = ok and own.evaluate([{kk: v for kk, v in ic[3].items() if kk != "model"}])[0] == ifull[3]
            if not ok:
                worst["model_tables"] = 1e9
                bad.append((tag, "a score depends on the models sharing the batch", 0.0, 0.0))
        except Exception as ex:      # noqa: BLE001
            bad.append((tag, "exception", repr(ex)[:200], 0.0))
    os.environ["AMPC_JIT"] = "0"
    # ---- SINDy libraries (incl. polynomial cross terms), both time modes, strict and true Jacobians ----
    from autompc_amd import SINDy
    from oracle.sindy import SINDyOracle
    worst.update({"sindy_pred": 0.0, "sindy_jac": 0.0})
    for case in range(max(6, n_cases // 10)):
        nx, nu = int(rng.integers(1, 6)), int(rng.integers(1, 3))
        tf = int(rng.integers(0, 4))
        inter = bool(tf > 0 and rng.random() < 0.5)
        pd = int(rng.integers(1, 5))
        cross = bool(pd > 1 and rng.random() < 0.6)
        if cross and nx + nu > 5 and pd > 3:
            pd = 3                                   # (keeps the library below the 4096-feature limit)
        mode = str(rng.choice(["discrete", "continuous"]))
        strict = bool(rng.random() < 0.5)
        prec = "f64" if rng.random() < 0.7 else "f32"
        tol = 1e-11 if prec == "f64" else 2e-4
        system = System(["x%d" % i for i in range(nx)], ["u%d" % i for i in range(nu)], dt=0.05)
        tag = "sindy case %d nx=%d nu=%d trig=%d inter=%d poly=%d cross=%d %s strict=%d %s" % (
            case, nx, nu, tf, inter, pd, cross, mode, strict, prec)
        try:
            m = SINDy(system, trig_basis=tf > 0, trig_freq=max(tf, 1), trig_interaction=inter, poly_basis=pd > 1,
                      poly_degree=pd, poly_cross_terms=cross, time_mode=mode, precision=prec, strict_reference=strict)
            nf = m.coefficients.shape[1]
            Xi = (rng.random((nx, nf)) < 0.3) * rng.normal(scale=0.2, size=(nx, nf))
            m.set_coefficients(Xi)
            orc = SINDyOracle(system, Xi, trig_freq=tf, trig_interaction=inter, poly_degree=pd, time_mode=mode,
                              strict_reference=strict, poly_cross_terms=cross)
            s_, c_ = rng.normal(scale=0.8, size=(37, nx)), rng.normal(scale=0.8, size=(37, nu))
            e = rel(m.pred_batch(s_, c_), orc.pred_batch(s_, c_))
            o_, jx, ju = m.pred_diff_batch(s_, c_)
            _, ojx, oju = orc.pred_diff_batch(s_, c_)
            ej = max(rel(jx, ojx), rel(ju, oju) if np.max(np.abs(oju)) > 0 else 0.0)
            worst["sindy_pred"] = max(worst["sindy_pred"], e / tol)
            worst["sindy_jac"] = max(worst["sindy_jac"], ej / (10 * tol))
            if e > tol or ej > 10 * tol:
                bad.append((tag, "sindy pred/jac", e, ej))
            # the MPPI rollout on this library (features over lanes, sindy_kernels.hpp) against the oracle's
            # MPPI on the oracle's model: one control step from the same numpy noise
            if prec == "f64" and mode == "discrete":
                from autompc_amd import MPPI as SMPPI, QuadCost as SQuadCost, Task as STask
                from oracle.costs import QuadCostOracle as SQuadCostOracle
                from oracle.mppi import MPPIOracle as SMPPIOracle
                Xs = Xi * 0.2                                  # (a contraction: the rollout stays finite)
                m.set_coefficients(Xs)
                orc2 = SINDyOracle(system, Xs, trig_freq=tf, trig_interaction=inter, poly_degree=pd, time_mode=mode,
                                   strict_reference=strict, poly_cross_terms=cross)
                S = rng.normal(size=(nx, nx))
                Qc = np.diag(rng.uniform(0.5, 2.0, size=nx)) + (0.05 * (S + S.T) if rng.random() < 0.5 else 0.0)
                Rc, Fc = np.diag(rng.uniform(0.01, 0.1, size=nu)), np.diag(rng.uniform(0.5, 2.0, size=nx))
                task = STask(system)
                task.set_cost(SQuadCost(system, Qc, Rc, Fc))
                task.set_ctrl_bounds(np.full(nu, -1.0), np.full(nu, 1.0))
                Hh, Np = int(rng.integers(3, 26)), int(rng.choice([64, 100, 256, 300]))
                sd = int(rng.integers(1 << 30))
                np.random.seed(sd)
                ctl = SMPPI(system, task, m, horizon=Hh, num_path=Np, sigma=0.25, lmda=1.0)
                np.random.seed(sd)
                oc = SMPPIOracle(orc2, SQuadCostOracle(Qc, Rc, Fc, np.zeros(nx)), np.tile([-1.0, 1.0], (nu, 1)),
                                horizon=Hh, num_path=Np, sigma=0.25, lmda=1.0)
                obs = rng.normal(scale=0.3, size=nx)
                cs = np.concatenate([obs, np.zeros(nu)])
                st = np.random.get_state()
                uo, _ = oc.run(cs, obs)
                np.random.set_state(st)
                uh, _ = ctl.run(cs, obs, return_details=True)
                em = max(rel(ctl.last_costs, oc.last_costs), rel(ctl.act_sequence, oc.act_sequence), rel(uh, uo))
                worst["sindy_mppi"] = max(worst.get("sindy_mppi", 0.0), em / 1e-9)
                if not em <= 1e-9:
                    bad.append((tag + " H=%d N=%d" % (Hh, Np), "sindy mppi", em, 0.0))
        except Exception as ex:      # noqa: BLE001
            bad.append((tag, "exception", repr(ex)[:200], 0.0))
    print("worst error / tolerance:", {k: float("%.3g" % v) for k, v in worst.items()})
    for b in bad:
        print("VIOLATION", b)
    print("%d cases (%d of them on run-time compiled kernels), %d violations" % (n_cases, n_jit[0], len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
