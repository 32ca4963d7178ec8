"""Host-side plugin surface on a CPU-only machine: construction, factories, state conventions,
pickling / deepcopy (the tuner pickles controllers, pipeline_tuner.py:216-218; Pipeline deep-copies
the task, pipeline.py:156) -- nothing here touches the GPU, device objects are created lazily."""
import copy
import pickle

import numpy as np
import pytest

from autompc_amd import System
from helpers import make_system


class _Cfg:
    """Stand-in for a ConfigSpace Configuration: factories only call get_dictionary()."""

    def __init__(self, **kw):
        self._d = kw

    def get_dictionary(self):
        return dict(self._d)


def _stack(nx=3, nu=2):
    from autompc_amd import MLP, QuadCost, Task
    system = make_system(nx, nu, dt=0.05)
    model = MLP(system, n_hidden_layers=2, hidden_size_1=32, hidden_size_2=48, nonlintype="tanh")
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(nx), 0.1 * np.eye(nu), np.eye(nx), goal=np.ones(nx)))
    task.set_ctrl_bounds(-np.ones(nu), 2 * np.ones(nu))
    task.set_num_steps(10)
    task.set_init_obs(np.zeros(nx))
    return system, task, model


def test_mppi_construction_matches_reference_conventions():
    from autompc_amd import MPPI, MPPIFactory, zeros
    system, task, model = _stack()
    np.random.seed(3)
    ref_draw = np.random.normal(scale=np.sqrt(0.5), size=(7, 2))
    np.random.seed(3)
    ctl = MPPIFactory(system)(_Cfg(horizon=7, sigma=0.5, lmda=0.3, num_path=50), task, model)
    assert isinstance(ctl, MPPI) and ctl.H == 7 and ctl.num_path == 50 and ctl.lmda == 0.3
    np.testing.assert_array_equal(ctl.act_sequence, ref_draw)      # random warm start, global stream
    np.testing.assert_array_equal(ctl.ctrl_scale, [2.0, 2.0])       # controls in units of umax
    assert ctl.state_dim == model.state_dim + system.ctrl_dim
    traj = zeros(system, 4)
    traj.obs[-1] = [1, 2, 3]
    traj.ctrls[-1] = [0.5, -0.5]
    np.testing.assert_array_equal(ctl.traj_to_state(traj), [1, 2, 3, 0.5, -0.5])
    first = ctl.act_sequence.copy()
    ctl.reset()                                                      # re-draws, like __init__
    assert ctl.act_sequence.shape == (7, 2) and not np.array_equal(ctl.act_sequence, first)
    assert ctl.step.__func__ is not None                             # newer-upstream alias of run()
    # factory kwargs override the configuration (controller.py:30-33)
    ctl2 = MPPIFactory(system, horizon=9)(_Cfg(horizon=7, num_path=20), task, model)
    assert ctl2.H == 9


def test_ilqr_construction_and_bounds_selection():
    from autompc_amd import IterativeLQR, IterativeLQRFactory, Task, QuadCost
    system, task, model = _stack()
    ctl = IterativeLQRFactory(system)(_Cfg(horizon=12), task, model)
    assert isinstance(ctl, IterativeLQR) and ctl.horizon == 12 and ctl.reuse_feedback == 0
    np.testing.assert_array_equal(ctl.ubounds[0], [-1, -1])          # bounded task -> clip
    free = Task(system)
    free.set_cost(QuadCost(system, np.eye(3), np.eye(2), np.eye(3)))
    assert IterativeLQR(system, free, model, 5).ubounds is None
    assert IterativeLQR(system, task, model, 5, reuse_feedback=99).reuse_feedback == 5
    with pytest.raises(NotImplementedError):
        IterativeLQR(system, task, model, 5, mode="barrier")
    with pytest.raises(Exception):
        IterativeLQR(system, task, model, 5, mode="bogus")


def test_controllers_and_models_pickle_and_deepcopy_without_device_state():
    from autompc_amd import MPPI, IterativeLQR
    system, task, model = _stack()
    np.random.seed(0)
    ctl = MPPI(system, task, model, horizon=6, num_path=30)
    for clone in (pickle.loads(pickle.dumps(ctl)), copy.deepcopy(ctl)):
        np.testing.assert_array_equal(clone.act_sequence, ctl.act_sequence)
        assert clone._plan is None and clone._handle is None and clone.H == 6
    il = pickle.loads(pickle.dumps(IterativeLQR(system, task, model, 8)))
    assert il.horizon == 8 and il._plan is None
    m2 = pickle.loads(pickle.dumps(model))
    assert m2._handle is None and all(np.array_equal(a, b) for a, b in zip(m2.weights, model.weights))
    t2 = copy.deepcopy(task)
    assert t2.get_cost().get_goal().tolist() == [1, 1, 1] and t2.get_num_steps() == 10


def test_mlp_parameter_dictionary_round_trip_uses_reference_keys():
    from autompc_amd import MLP
    system, _, model = _stack()
    params = model.get_parameters()
    assert sorted(params) == ["dy_means", "dy_std", "net_state", "xu_means", "xu_std"]
    assert sorted(params["net_state"]) == ["layers.layer0.bias", "layers.layer0.weight",
                                           "layers.layer1.bias", "layers.layer1.weight",
                                           "output_layer.bias", "output_layer.weight"]
    other = MLP(system, n_hidden_layers=2, hidden_size_1=32, hidden_size_2=48, nonlintype="tanh", seed=7)
    assert not np.array_equal(other.weights[0], model.weights[0])
    other.set_parameters(params)
    assert all(np.array_equal(a, b) for a, b in zip(other.weights, model.weights))
    bad = MLP(system, n_hidden_layers=1, hidden_size_1=16)
    with pytest.raises((ValueError, KeyError)):
        bad.set_parameters(params)
    assert model.is_diff and not model.is_linear and model.state_dim == 3


def test_non_quadratic_costs_and_foreign_models_are_rejected_loudly():
    from autompc_amd import MPPI, Task, ThresholdCost
    system, task, model = _stack()

    class Foreign:                      # a model without device staging: no silent CPU path
        state_dim = 3
        system = None
    with pytest.raises(TypeError):
        MPPI(system, task, Foreign(), horizon=5)
    t = Task(system)
    t.set_cost(ThresholdCost(system, np.zeros(3), (0, 2), 0.1))
    t.set_ctrl_bounds(-np.ones(2), np.ones(2))
    ctl = MPPI(system, t, model, horizon=5, num_path=10)
    from autompc_amd import _lib
    with pytest.raises((TypeError, _lib.AmpcError)):     # AmpcError: no GPU here; TypeError on one
        ctl.run(np.zeros(5), np.zeros(3))


def test_simulate_drives_any_controller_like_the_reference_driver():
    from autompc_amd import simulate, Controller

    class Const(Controller):
        def __init__(self, system):
            super().__init__(system, None, None)

        def traj_to_state(self, traj):
            return np.concatenate([traj[-1].obs, traj[-1].ctrl])

        def run(self, state, new_obs):
            u = np.array([0.5, -0.25])
            return u, np.concatenate([new_obs, u])

        @property
        def state_dim(self):
            return 5
    system = make_system(3, 2)
    traj = simulate(Const(system), np.array([1.0, 0.0, 0.0]), dynamics=lambda x, u: x + 0.1 * u.sum(),
                    max_steps=4)
    assert len(traj) == 5 and np.allclose(traj.ctrls[:-1], [0.5, -0.25]) and np.all(traj.ctrls[-1] == 0)
    np.testing.assert_allclose(traj.obs[-1], [1.1, 0.1, 0.1])
    with pytest.raises(ValueError):
        simulate(Const(system), np.zeros(3))


def test_mlp_training_fits_a_linear_system_on_the_host():
    """MLP.train (PyTorch, mlp.py:177-217: Adam + SmoothL1 on normalised deltas) produces weights
    and normalisers that predict a simple system; checked with the oracle's forward pass, so no
    GPU is needed.  (Training is outside the MPC inner loop and is not a HIP kernel.)"""
    from autompc_amd import MLP, zeros
    from oracle import mlp as omlp
    system = make_system(2, 1)
    A = np.array([[0.95, 0.1], [-0.1, 0.9]])
    Bm = np.array([[0.0], [0.2]])
    rng = np.random.default_rng(0)
    trajs = []
    for _ in range(8):
        tr = zeros(system, 40)
        x = rng.uniform(-1, 1, size=2)
        for t in range(40):
            u = rng.uniform(-1, 1, size=1)
            tr.obs[t], tr.ctrls[t] = x, u
            x = A @ x + Bm @ u
        trajs.append(tr)
    m = MLP(system, n_hidden_layers=2, hidden_size=32, nonlintype="tanh", n_train_iters=60,
            n_batch=32, lr=5e-3)
    before = [w.copy() for w in m.weights]
    m.train(trajs)
    assert any(not np.array_equal(a, b) for a, b in zip(before, m.weights))
    p = omlp.make_params(m.weights, m.biases, "tanh", m.xu_means, m.xu_std, m.dy_means, m.dy_std)
    s = rng.uniform(-1, 1, size=(64, 2))
    c = rng.uniform(-1, 1, size=(64, 1))
    pred = omlp.pred_batch(p, s, c)
    truth = s @ A.T + c @ Bm.T
    assert np.sqrt(np.mean((pred - truth) ** 2)) < 0.05
    keys = m.get_parameters()
    assert set(keys) == {"net_state", "xu_means", "xu_std", "dy_means", "dy_std"}


# ---- episode semantics of the tuner's objective (pipeline_tuner.py:222-231) ---------------------
class _ReferenceStyleTask:
    """The termination-condition slot exactly as the reference's Task keeps it
    (tasks/task.py:39-53, 73-101): set_num_steps installs a closure over num_steps."""

    def __init__(self):
        self._term_cond = None
        self._num_steps = None

    def set_num_steps(self, num_steps):
        self._term_cond = lambda traj: len(traj) >= num_steps
        self._num_steps = num_steps

    def has_num_steps(self):
        return self._num_steps is not None

    def get_num_steps(self):
        return self._num_steps

    def term_cond(self, traj):
        return self._term_cond(traj) if self._term_cond is not None else False

    def set_term_cond(self, term_cond):
        self._term_cond = term_cond


def _host_simulate_rows(task):
    """Rows simulate() returns for the task's own termination condition (utils/simulation.py:52-64)."""
    rows = 1
    for _ in range(task.get_num_steps() if task.has_num_steps() else 10000):
        rows += 1
        if task.term_cond([None] * rows):
            break
    return rows


def test_default_episode_is_num_steps_rows():
    from autompc_amd import Task
    from autompc_amd.tuning.batch_eval import default_episode_controls, episode_of
    system = System(["x"], ["u"])
    for make in (lambda: Task(system), _ReferenceStyleTask):
        for n in (0, 1, 2, 7, 200):
            task = make()
            task.set_num_steps(n)
            max_steps, tc = episode_of(task)
            assert max_steps == n and tc is None
            assert default_episode_controls(task) + 1 == _host_simulate_rows(task)
        assert default_episode_controls(make()) == 10000
    task = Task(system)
    task.set_num_steps(7)
    assert default_episode_controls(task) == 6          # 7 rows, 6 controls


def test_user_term_cond_is_recognised():
    from autompc_amd import Task
    from autompc_amd.tuning.batch_eval import episode_of
    system = System(["x"], ["u"])
    for make in (lambda: Task(system), _ReferenceStyleTask):
        task = make()
        task.set_num_steps(30)
        task.set_term_cond(lambda traj: len(traj) >= 5)
        max_steps, tc = episode_of(task)
        assert max_steps == 30 and tc is not None and tc([0] * 5) and not tc([0] * 4)
        task.set_num_steps(12)                          # a later set_num_steps replaces it again
        assert episode_of(task) == (12, None)
        other = make()
        other.set_term_cond(lambda traj: len(traj) >= 3)
        max_steps, tc = episode_of(other)
        assert max_steps == 10000 and tc is not None
    # a closure over a different count than the task's is not the default condition
    odd = _ReferenceStyleTask()
    odd.set_num_steps(9)
    odd._num_steps = 20
    assert episode_of(odd)[1] is not None


def test_evaluator_rejects_horizon_above_cap():
    from autompc_amd.tuning import CandidateEvaluator

    class _Stub:
        state_dim = 1

        def stage_into(self, h):
            raise AssertionError("must fail before touching the device")
    from autompc_amd import QuadCost, Task
    system = System(["x"], ["u"])
    task = Task(system)
    task.set_cost(QuadCost(system, np.eye(1), np.eye(1), np.eye(1)))
    task.set_ctrl_bounds([-1.0], [1.0])
    ev = CandidateEvaluator(system, task, _Stub(), horizon_cap=20)
    cand = dict(horizon=21, sigma=1.0, lmda=1.0, num_path=16, Q=[1.0], R=[1.0], F=[1.0])
    with pytest.raises(ValueError, match="horizon_cap"):
        ev.evaluate([cand])


def test_ilqr_traj_to_state_default_is_the_references_bare_model_state():
    """ilqr.py:96-98 returns model.traj_to_state(traj); run() strips ctrl_dim entries off it anyway
    (:278).  Default = that, for every model whose update_state ignores the old state; ARX
    (update_state shifts the old state, arx.py:113-127) gets model state + last control -- the only
    layout it can be simulated with; strict_reference forces either."""
    from autompc_amd import ARX, IterativeLQR, zeros
    system, task, model = _stack()
    traj = zeros(system, 3)
    traj.obs[:] = np.arange(9.0).reshape(3, 3)
    traj.ctrls[:] = [[0.1, 0.2], [0.3, 0.4], [0.5, 0.6]]
    np.testing.assert_array_equal(IterativeLQR(system, task, model, 5).traj_to_state(traj), [6, 7, 8])
    np.testing.assert_array_equal(IterativeLQR(system, task, model, 5, strict_reference=True).traj_to_state(traj), [6, 7, 8])
    np.testing.assert_array_equal(IterativeLQR(system, task, model, 5, strict_reference=False).traj_to_state(traj),
                                  [6, 7, 8, 0.5, 0.6])
    arx = ARX(system, history=2)
    arx.A, arx.B = np.zeros((arx.state_dim, arx.state_dim)), np.zeros((arx.state_dim, 2))
    auto = IterativeLQR(system, task, arx, 5).traj_to_state(traj)
    assert auto.shape == (arx.state_dim + 2,) and list(auto[-2:]) == [0.5, 0.6]
    bare = IterativeLQR(system, task, arx, 5, strict_reference=True).traj_to_state(traj)
    np.testing.assert_array_equal(bare, auto[:-2])
