"""SURVEY 8 f4 on the device: fit with PyTorch-ROCm -> stage from device memory (ampc_set_mlp_dev) -> predict
with the HIP kernels.  ``-m gpu``."""
import time

import numpy as np
import pytest
import torch

from autompc_amd import MLP, _lib, zeros
from autompc_amd.sysid import mlp_fit as F
from autompc_amd.sysid.mlp import MLPFactory
from oracle import mlp as omlp
from helpers import make_system
from test_mlp_fit import _case, _interleave

pytestmark = pytest.mark.gpu


def _torch_forward(ws, bs, act, norms, states, ctrls):
    """x' = x + dy_means + dy_std * net(([x, u] - xu_means) / xu_std) in torch float64 on the parameters' device."""
    dev = ws[0].device
    xm, xs, dm, ds = [torch.as_tensor(v, dtype=torch.float64, device=dev) for v in norms]
    x = torch.as_tensor(states, dtype=torch.float64, device=dev)
    h = (torch.cat([x, torch.as_tensor(ctrls, dtype=torch.float64, device=dev)], dim=1) - xm) / xs
    f = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "selu": torch.selu}[act]
    for w, b in zip(ws[:-1], bs[:-1]):
        h = f(h @ w.T + b)
    return (x + dm + ds * (h @ ws[-1].T + bs[-1])).cpu().numpy()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("nx,nu,hidden,act", [(17, 6, [256, 256], "relu"), (2, 1, [64, 64], "tanh"),
                                             (4, 2, [100], "sigmoid"), (3, 2, [48, 200, 17, 64], "selu"),
                                             (40, 3, [192, 64], "relu")])
def test_a_model_staged_from_device_memory_is_the_host_staged_model_bit_for_bit(nx, nu, hidden, act, precision):
    """ampc_set_mlp_dev folds the normalisers and writes every fragment packing on the device; predictions and
    Jacobians of the two handles must be IDENTICAL (same packed bytes), in both precisions."""
    p = omlp.random_params(nx, nu, hidden, act, seed=3)
    rng = np.random.default_rng(1)
    p["xu_means"], p["xu_std"] = rng.normal(scale=0.3, size=nx + nu), rng.uniform(0.5, 2.0, size=nx + nu)
    p["dy_means"], p["dy_std"] = rng.normal(scale=0.02, size=nx), rng.uniform(0.05, 0.2, size=nx)
    host, dev = _lib.Handle(0, precision), _lib.Handle(0, precision)
    host.set_mlp(nx, nu, p["weights"], p["biases"], act, p["xu_means"], p["xu_std"], p["dy_means"], p["dy_std"])
    tw = [torch.from_numpy(np.ascontiguousarray(w)).cuda() for w in p["weights"]]
    tb = [torch.from_numpy(np.ascontiguousarray(b)).cuda() for b in p["biases"]]
    tn = [torch.from_numpy(np.ascontiguousarray(p[k])).cuda() for k in ("xu_means", "xu_std", "dy_means", "dy_std")]
    torch.cuda.synchronize()
    dev.set_mlp_dev(nx, nu, hidden, [t.data_ptr() for t in tw], [t.data_ptr() for t in tb], act,
                    [t.data_ptr() for t in tn])
    del tw, tb, tn                                # the arrays are only read during the call
    s, c = rng.normal(size=(96, nx)), rng.normal(size=(96, nu))
    np.testing.assert_array_equal(dev.pred_batch(s, c), host.pred_batch(s, c))
    if precision == "f64":
        for a, b in zip(dev.pred_diff_batch(s[:8], c[:8]), host.pred_diff_batch(s[:8], c[:8])):
            np.testing.assert_array_equal(a, b)
    host.close()
    dev.close()


def test_set_mlp_dev_refuses_host_pointers():
    h = _lib.Handle(0, "f64")
    w = [np.zeros((16, 3)), np.zeros((2, 16))]
    b = [np.zeros(16), np.zeros(2)]
    n = [np.zeros(3), np.ones(3), np.zeros(2), np.ones(2)]
    with pytest.raises(_lib.AmpcError, match="device memory"):
        h.set_mlp_dev(2, 1, [16], [a.ctypes.data for a in w], [a.ctypes.data for a in b], "relu",
                      [a.ctypes.data for a in n])
    h.close()


@pytest.mark.parametrize("tag", ["p_tanh", "hc_relu3", "p_selu1"])
def test_train_on_the_gpu_stage_and_predict(tag):
    """MLP.train on cuda (HIP-graph-captured lockstep fit) reproduces the REFERENCE's trained net (its CPU
    training, tests/golden/mlpfit_*) to 1e-9; the staged model's pred_batch equals the same parameters' torch
    float64 forward pass to 1e-12 and the reference's own predictions to 1e-8."""
    g, system, trajs, hidden, _, final = _case(tag)
    act = str(g["activation"])
    kw = {"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)}
    m = MLP(system, n_hidden_layers=len(hidden), nonlintype=act, n_train_iters=int(g["n_train_iters"]),
            n_batch=int(g["n_batch"]), lr=float(g["lr"]), seed=int(g["seed"]), **kw)
    m.train(trajs)
    dp = m._dev_params
    assert dp is not None and dp["w"][0].is_cuda and m._weights is None      # parameters stayed on the device
    pred = m.pred_batch(g["states"], g["ctrls_q"])                          # staged through ampc_set_mlp_dev
    assert m._weights is None                                               # ... without a host copy
    want = _torch_forward(dp["w"], dp["b"], act, dp["norm_np"], g["states"], g["ctrls_q"])
    np.testing.assert_allclose(pred, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(pred, g["pred"], rtol=0, atol=1e-8)
    for a, b in zip(_interleave(m.weights, m.biases), final):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
    # eager (no graphs) and graph-captured steps are the same program
    fit_args = ([[system.obs_dim + system.ctrl_dim] + hidden + [system.obs_dim]], act, [float(g["lr"])], [int(g["seed"])])
    XU, dY, xm, xs, dm, ds = F.training_arrays(trajs)
    feed, target = [torch.from_numpy(v) for v in F.normalised(XU, dY, xm, xs, dm, ds)]
    eager = F.LockstepFit(*fit_args, feed, target, int(g["n_batch"]), device="cuda", use_graphs=False)
    eager.run(int(g["n_train_iters"]))
    for a, b in zip(_interleave(*[[t.cpu().numpy() for t in part] for part in eager.parameters(0)]),
                    _interleave(m.weights, m.biases)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


def _halfcheetah_like_trajs(system, n_traj, rows, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_traj):
        t = zeros(system, rows)
        t.obs[:] = 0.05 * rng.normal(size=(rows, system.obs_dim)).cumsum(axis=0)
        t.ctrls[:] = rng.uniform(-1, 1, size=(rows, system.ctrl_dim))
        out.append(t)
    return out


def test_eight_2x256_models_fit_in_lockstep_at_five_times_the_sequential_rate():
    """K = 8 HalfCheetah-shaped 2 x 256 models, own seeds and learning rates, 3 epochs over 3200 rows: the
    lockstep fit agrees with each model's own reference-style fit on the GPU (1e-7: 150 Adam steps of bmm /
    mm summation-order differences) and, once its graphs are captured (the first epoch; a fit is 50), runs at more
    than five times the rate of the eight sequential eager fits."""
    system = make_system(17, 6)
    trajs = _halfcheetah_like_trajs(system, 16, 201)
    XU, dY, xm, xs, dm, ds = F.training_arrays(trajs)
    feed, target = [torch.from_numpy(v).cuda() for v in F.normalised(XU, dY, xm, xs, dm, ds)]
    dims = [[23, 256, 256, 17]] * 8
    lrs = [1e-3 * (1 + k) for k in range(8)]
    seeds = list(range(20, 28))
    F.fit_reference_style(dims[0], "relu", feed, target, 1, 64, 1e-3, 1, device="cuda")     # warm the libraries
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    seq = [F.fit_reference_style(dims[k], "relu", feed, target, 3, 64, lrs[k], seeds[k], device="cuda") for k in range(8)]
    torch.cuda.synchronize()
    t_seq = (time.perf_counter() - t0) * 2.0 / 3.0                  # two epochs' worth
    fit = F.LockstepFit(dims, "relu", lrs, seeds, feed, target, 64, device="cuda")
    fit.run(1)                                                      # (captures the chunk graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit.run(2)
    torch.cuda.synchronize()
    t_lock = time.perf_counter() - t0
    for k in range(8):
        lw, lb = fit.parameters(k)
        for a, b in zip(lw + lb, seq[k][0] + seq[k][1]):
            assert float((a - b).abs().max()) < 1e-7
    print("two epochs: sequential %.3f s, lockstep %.3f s: %.1fx" % (t_seq, t_lock, t_seq / t_lock))
    assert t_seq / t_lock >= 5.0


def test_tuner_model_axis_on_the_device():
    """BatchPipelineTuner(model_factory=..., trajs=...) over the real CandidateEvaluator: every configuration's
    MLP is fitted on the GPU and its candidate is scored on THAT model -- the same score an evaluator run gives
    when handed a model trained by its own MLP.train(trajs) (staged from the host arrays)."""
    from autompc_amd import QuadCost, Task
    from autompc_amd.tuning import (BatchPipelineTuner, CandidateEvaluator, DictConfiguration,
                                    candidate_from_config, sample_pipeline_configs)
    system = make_system(3, 2)
    trajs = _halfcheetah_like_trajs(system, 4, 60, seed=3)
    surrogate = MLP(system, n_hidden_layers=2, hidden_size=32, nonlintype="tanh", n_train_iters=2, n_batch=32)
    surrogate.train(trajs)
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(3), 0.1 * np.eye(2), np.eye(3)))
    task.set_ctrl_bounds(-np.ones(2), np.ones(2))
    task.set_num_steps(12)
    task.set_init_obs(np.array([0.3, -0.2, 0.1]))
    rng = np.random.default_rng(4)
    cfgs = sample_pipeline_configs(system, 10, rng, model_axis=True)
    for c in cfgs:                                       # cost gains that keep a 12-step episode finite
        for k in c:
            if k.startswith("_cost:"):
                c[k] = float(c[k] ** 0.25)
        c["_ctrlr:num_path"] = 128
    factory = MLPFactory(system, n_train_iters=2, n_batch=32)
    ev = CandidateEvaluator(system, task, surrogate)
    tuner = BatchPipelineTuner(system, ev, batch_size=5, model_factory=factory, trajs=trajs)
    best, res = tuner.run(10, np.random.default_rng(0), seed=7, configs=cfgs)
    assert tuner.models_fitted == 10 and np.all(np.isfinite(res.costs))
    assert best is cfgs[int(np.argmin(res.costs))]
    for i in (0, 4, 9):
        cand = candidate_from_config(system, cfgs[i])
        m = factory(DictConfiguration(cand["model_cfg"]), trajs)          # its own train()
        m.weights = [w.copy() for w in m.weights]                         # (host arrays: staged by ampc_set_mlp)
        cand["model"] = m
        want = ev.evaluate([cand], seed=7, index_offset=i)[0]
        assert abs(res.costs[i] - want) <= 1e-9 * max(1.0, abs(want))
