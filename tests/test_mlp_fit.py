"""MLP fitting against the reference's own ``MLP(...).train(trajs)``.

``tests/golden/mlpfit_*.npz`` hold, from the imported reference (gen_golden.gen_mlpfit): the training
trajectories, the net as constructed, the normalisers and the net after training, and its predictions.
``sysid/mlp_fit.py`` must reproduce them: the reference-style loop (nn.Linear + torch.optim.Adam), the lockstep
fit (stacked parameters, hand-written backward pass, flat Adam) alone and with several models of different
widths side by side, and ``MLP.train`` itself.  CPU: torch trains here without a GPU (inference does not).

Tolerance: the initial weights and the normalisers are bit for bit the reference's; fitted weights agree to
1e-10 (same mini-batch order, same arithmetic; bmm / mm summation order and a BLAS build may differ in the last
bits, and Adam divides by sqrt(v) + 1e-8).
"""
import numpy as np
import pytest
import torch

from autompc_amd import MLP, zeros
from autompc_amd.sysid import mlp_fit as F
from oracle import mlp as omlp
from conftest import golden
from helpers import make_system

CASES = ["p_tanh", "hc_relu3", "p_selu1"]
TOL = 1e-10


def _case(tag):
    g = golden("mlpfit_" + tag)
    nx, nu = int(g["nx"]), int(g["nu"])
    system = make_system(nx, nu)
    trajs = []
    for o, c in zip(g["obs"], g["ctrls"]):
        t = zeros(system, o.shape[0])
        t.obs[:], t.ctrls[:] = o, c
        trajs.append(t)
    hidden = [int(h) for h in g["hidden"]]
    n_lin = len(hidden) + 1
    init = [g["init_%d" % i] for i in range(2 * n_lin)]
    final = [g["final_%d" % i] for i in range(2 * n_lin)]
    return g, system, trajs, hidden, init, final


def _interleave(ws, bs):
    return [np.asarray(x) for pair in zip(ws, bs) for x in pair]


@pytest.mark.parametrize("tag", CASES)
def test_initial_weights_and_normalisers_are_the_references_bit_for_bit(tag):
    g, system, trajs, hidden, init, _ = _case(tag)
    dims = [system.obs_dim + system.ctrl_dim] + hidden + [system.obs_dim]
    ws, bs = F.initial_parameters(int(g["seed"]), dims)
    for a, b in zip(_interleave([w.numpy() for w in ws], [b.numpy() for b in bs]), init):
        np.testing.assert_array_equal(a, b)
    _, _, xm, xs, dm, ds = F.training_arrays(trajs)
    for a, key in ((xm, "xu_means"), (xs, "xu_std"), (dm, "dy_means"), (ds, "dy_std")):
        np.testing.assert_array_equal(a, g[key])


@pytest.mark.parametrize("tag", CASES)
def test_reference_style_and_lockstep_fits_reproduce_the_references_trained_net(tag):
    g, system, trajs, hidden, _, final = _case(tag)
    dims = [system.obs_dim + system.ctrl_dim] + hidden + [system.obs_dim]
    XU, dY, xm, xs, dm, ds = F.training_arrays(trajs)
    feed, target = [torch.from_numpy(v) for v in F.normalised(XU, dY, xm, xs, dm, ds)]
    act, lr, seed = str(g["activation"]), float(g["lr"]), int(g["seed"])
    n_iter, n_batch = int(g["n_train_iters"]), int(g["n_batch"])
    rw, rb = F.fit_reference_style(dims, act, feed, target, n_iter, n_batch, lr, seed)
    for a, b in zip(_interleave([w.numpy() for w in rw], [b.numpy() for b in rb]), final):
        np.testing.assert_allclose(a, b, rtol=0, atol=TOL)
    fit = F.LockstepFit([dims], act, [lr], [seed], feed, target, n_batch, device="cpu")
    fit.run(n_iter)
    lw, lb = fit.parameters(0)
    for a, b in zip(_interleave([w.numpy() for w in lw], [b.numpy() for b in lb]), final):
        np.testing.assert_allclose(a, b, rtol=0, atol=TOL)
    assert fit.steps_done == n_iter * -(-feed.shape[0] // n_batch)


@pytest.mark.parametrize("tag", CASES)
def test_mlp_train_is_the_references_train(tag):
    """MLP(...).train(trajs) as a user calls it; the fitted model's prediction (oracle forward pass on the
    fitted parameters: no GPU here) equals the reference's pred_batch after ITS training."""
    g, system, trajs, hidden, _, final = _case(tag)
    kw = {"hidden_size_%d" % (i + 1): h for i, h in enumerate(hidden)}
    m = MLP(system, n_hidden_layers=len(hidden), nonlintype=str(g["activation"]), n_train_iters=int(g["n_train_iters"]),
            n_batch=int(g["n_batch"]), lr=float(g["lr"]), seed=int(g["seed"]), **kw)
    m.train(trajs)
    for a, b in zip(_interleave(m.weights, m.biases), final):
        np.testing.assert_allclose(a, b, rtol=0, atol=TOL)
    np.testing.assert_array_equal(m.xu_std, g["xu_std"])
    p = omlp.make_params(m.weights, m.biases, str(g["activation"]), m.xu_means, m.xu_std, m.dy_means, m.dy_std)
    np.testing.assert_allclose(omlp.pred_batch(p, g["states"], g["ctrls_q"]), g["pred"], rtol=0, atol=1e-9)
    # the reference's parameter dictionary round-trips the fitted net (mlp.py:308-321)
    m2 = MLP(system, n_hidden_layers=len(hidden), nonlintype=str(g["activation"]), **kw)
    m2.set_parameters(m.get_parameters())
    for a, b in zip(_interleave(m2.weights, m2.biases), _interleave(m.weights, m.biases)):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("act", F.ACTS)
def test_lockstep_models_of_different_widths_equal_their_sequential_fits(act):
    """Three models (own widths, seeds and learning rates) zero-padded into one stacked program: each ends
    where its own reference-style fit ends, and the padding stays exactly zero."""
    g, system, trajs, _, _, _ = _case("p_tanh")
    XU, dY, xm, xs, dm, ds = F.training_arrays(trajs)
    feed, target = [torch.from_numpy(v) for v in F.normalised(XU, dY, xm, xs, dm, ds)]
    dims = [[5, 32, 24, 3], [5, 17, 40, 3], [5, 32, 40, 3]]
    lrs, seeds = [3e-3, 1e-3, 1e-2], [7, 8, 9]
    fit = F.LockstepFit(dims, act, lrs, seeds, feed, target, 64, device="cpu")
    assert fit.padded
    fit.run(3)
    for k in range(3):
        rw, rb = F.fit_reference_style(dims[k], act, feed, target, 3, 64, lrs[k], seeds[k])
        lw, lb = fit.parameters(k)
        for a, b in zip(lw + lb, rw + rb):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=TOL)
    assert float((fit.flat * (1.0 - fit.mask)).abs().max()) == 0.0
    # a second run() continues the same optimisation (Adam's step count carries on)
    fit.run(1)
    rw, rb = F.fit_reference_style(dims[1], act, feed, target, 4, 64, lrs[1], seeds[1])
    lw, lb = fit.parameters(1)
    for a, b in zip(lw + lb, rw + rb):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=TOL)


def test_fit_mlps_groups_by_depth_and_activation_and_fits_every_model_as_its_own_train_would():
    g, system, trajs, _, _, _ = _case("p_tanh")
    specs = [(2, "tanh", (32, 24), 1e-3, 1), (2, "tanh", (20, 20), 5e-3, 2), (1, "relu", (30,), 1e-3, 3),
             (2, "relu", (16, 16), 1e-3, 4)]

    def build():
        return [MLP(system, n_hidden_layers=d, nonlintype=a, n_train_iters=2, n_batch=64, lr=lr, seed=s,
                    **{"hidden_size_%d" % (i + 1): h for i, h in enumerate(hs)}) for d, a, hs, lr, s in specs]
    together, alone = build(), build()
    info = F.fit_mlps(together, trajs)
    assert info["groups"] == 3 and info["steps"] == 2 * 3
    for m in alone:
        m.train(trajs)
    for a, b in zip(together, alone):
        for x, y in zip(_interleave(a.weights, a.biases), _interleave(b.weights, b.biases)):
            np.testing.assert_allclose(x, y, rtol=0, atol=TOL)
        np.testing.assert_array_equal(a.dy_std, b.dy_std)


def test_a_fitted_model_survives_deepcopy_and_assigning_weights_drops_the_device_copy():
    import copy
    g, system, trajs, hidden, _, _ = _case("p_selu1")
    m = MLP(system, n_hidden_layers=1, hidden_size_1=hidden[0], nonlintype="selu", n_train_iters=1, n_batch=32)
    m.train(trajs)
    assert m._dev_params is not None
    c = copy.deepcopy(m)
    assert c._dev_params is None
    for a, b in zip(_interleave(c.weights, c.biases), _interleave(m.weights, m.biases)):
        np.testing.assert_array_equal(a, b)
    m.weights = [w * 2.0 for w in m.weights]
    assert m._dev_params is None and np.array_equal(m.biases[0], c.biases[0])
