"""Iterative LQR controller whose solve runs on MI355X.

Drop-in for the reference's ``autompc.control.IterativeLQR`` / ``IterativeLQRFactory``
(reference: autompc/control/ilqr.py:16-41 factory, :43-295 controller): same constructor
(``horizon``, ``reuse_feedback``, ``ubounds``, ``mode``, ``verbose``), same
``compute_ilqr_default`` return tuple ``(converged, states, ctrls, Ks, ks)``, same ``run`` (full
re-solve from a zero guess every control step, ``u = u0 + K0 (x - x0)``).

Everything numerical happens in ``ampc_ilqr_solve`` (csrc/ilqr_kernels.hpp): rollout of the
guess, backward Riccati sweeps with an unregularised partial-pivot LU, the 10-step-size batched
line search through the MFMA MLP tile, acceptance logic, and the analytic Jacobian refresh.
A singular ``Quu`` surfaces as ``numpy.linalg.LinAlgError`` exactly where the reference's
``np.linalg.solve`` would raise it (the tuner catches that, pipeline_tuner.py:236-239).

Deviations, on purpose: ``mode='barrier'|'auglag'`` raise NotImplementedError at construction
(the reference binds them to methods that do not exist, ilqr.py:71-74); ``state_dim`` returns
``model.state_dim + ctrl_dim`` (the reference's property references an undefined name,
ilqr.py:84-87).

``traj_to_state`` (``strict_reference``).  The reference returns the BARE model state
(ilqr.py:96-98) although ``run()`` strips ``ctrl_dim`` entries off whatever it is handed (:278-279)
and returns model state + control.  That is harmless for models whose ``update_state`` ignores the
old state (MLP, SINDy, Koopman) and a shape error for ARX, whose ``update_state`` shifts the old
state (arx.py:113-127).  ``strict_reference=True`` reproduces the reference for every model;
``False`` always returns model state + last control (the layout ``run()`` consumes);
the default ``None`` is the reference's bare state unless the model declares
``update_state_reads_state`` (ARX), where only the second form can be simulated at all.

Precision.  The solve is f64, like the reference.  A full iLQR solve is a chain of up to 50
discrete line-search / acceptance / convergence decisions (``ratio > 0.3``, ``||du|| < 1e-3``,
ilqr.py:207-263); in f32 one of them eventually falls the other way and the solve then stops at
a different -- equally converged -- iterate: measured against the reference's golden solves the
f32 states deviate by up to 4e-3 relative (tests/test_gpu_ilqr.py), outside north_star's 1e-4.
``precision="f32"`` is therefore refused unless ``allow_inexact=True`` is passed as well.
"""
import numpy as np

from .. import _lib
from .controller import Controller, ControllerFactory
from .mppi import _stage_cost
from ..costs.blocks import is_quad_sum


class IterativeLQR(Controller):
    def __init__(self, system, task, model, horizon, reuse_feedback=-1, ubounds=None, mode=None,
                 verbose=False, precision=None, device=None, allow_inexact=False, strict_reference=None):
        super().__init__(system, task, model)
        if not hasattr(model, "stage_into"):
            raise TypeError("IterativeLQR needs a device-stageable model (autompc_amd.sysid.MLP); "
                            "there is no CPU fallback")
        self.horizon = int(horizon)
        self.dt = system.dt
        if reuse_feedback is None or reuse_feedback <= 0:
            self.reuse_feedback = 0
        else:
            self.reuse_feedback = min(int(reuse_feedback), self.horizon)
        if ubounds is None and task.are_ctrl_bounded():
            b = task.get_ctrl_bounds()
            self.ubounds = (b[:, 0], b[:, 1])
        else:
            self.ubounds = ubounds
        if mode is not None:
            if mode in ("barrier", "auglag"):
                raise NotImplementedError("mode=%r is not implemented (nor in the reference)" % mode)
            raise Exception("mode has to be None/barrier/auglag")
        self.mode = mode
        self.verbose = verbose
        self.precision = precision or getattr(model, "precision", "f64")
        if self.precision != "f64" and not allow_inexact:
            raise ValueError("IterativeLQR solves in f64: an f32 solve takes different line-search / "
                             "convergence decisions than the reference and ends up to 4e-3 away from "
                             "its result (parity tolerance 1e-4).  Pass allow_inexact=True to run the "
                             "f32 kernels anyway.")
        self.allow_inexact = bool(allow_inexact)
        self.strict_reference = strict_reference
        self.device = device if device is not None else getattr(model, "device", 0)
        self.compute_ilqr = self.compute_ilqr_default
        self._handle = self._plan = None
        self._terminal_goal = False
        self._jit_pending = False
        self.reset()

    def reset(self):
        self._need_recompute = True
        self._step_count = 0
        self._states = None
        self._guess = None

    def _device(self):
        if self._plan is not None and self._jit_pending:
            # the kernels specialised for this model's shape were still compiling when the plan
            # was made (csrc/jit_host.hpp): switch over once they are ready (same results)
            st = self._handle.jit_status()[0]
            if st == 2:
                self._plan.close()
                self._plan = None
            self._jit_pending = st == 1
        bounded = self.ubounds is not None
        if self._handle is None:
            h = _lib.Handle(self.device, self.precision)
            self.model.stage_into(h)
            blk = _stage_cost(h, self.task.get_cost(), self.system.obs_dim, self.system.ctrl_dim)
            # QuadCost(strict_reference=False) opts out of the reference's goal-less terminal
            # gradient (cost.py:195): the device sweep then seeds v_N = (F+F')(x_N - goal) + lin_term
            self._terminal_goal = blk["terminal_goal"]
            if bounded:
                h.set_ctrl_bounds(np.asarray(self.ubounds[0], dtype=float),
                                  np.asarray(self.ubounds[1], dtype=float))
            self._handle = h
        if self._plan is None:
            self._plan = _lib.IlqrPlan(self._handle, 1, self.horizon, self.dt, clip_to_bounds=bounded,
                                       terminal_goal=self._terminal_goal)
            self._jit_pending = self._plan.kernel_kind() == 0 and self._handle.jit_status()[0] == 1
        return self._plan

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_handle"] = state["_plan"] = None
        return state

    def compute_ilqr_default(self, state, uguess, u_threshold=1e-3, max_iter=50, ls_max_iter=10,
                             ls_discount=0.2, ls_cost_threshold=0.3, silent=False):
        consts = (float(u_threshold), int(ls_max_iter), float(ls_discount), float(ls_cost_threshold))
        plan = self._device()
        if consts != getattr(plan, "_consts", (1e-3, 10, 0.2, 0.3)):
            # the reference takes them per call (ilqr.py:100-101): they are kernel arguments of the plan
            if not 1 <= consts[1] <= 16:
                raise ValueError("ls_max_iter must be in 1..16 (the step sizes are the rows of one MFMA tile)")
            plan.set_constants(*consts)
            plan._consts = consts
        out = plan.solve(np.asarray(state)[None, :], np.asarray(uguess)[None], max_iter)
        if out["status"][0] == 1:
            raise np.linalg.LinAlgError("Singular matrix")
        self.last_iters = int(out["iters"][0])
        self.last_objective = float(out["objective"][0])
        return (bool(out["converged"][0]), out["states"][0], out["ctrls"][0], out["Ks"][0],
                out["ks"][0])

    def run(self, constate, new_obs, silent=True):
        nu = self.system.ctrl_dim
        state = self.model.update_state(constate[:-nu], constate[-nu:], new_obs)
        if self._need_recompute:
            converged, states, ctrls, Ks, ks = self.compute_ilqr(
                state, np.zeros((self.horizon, nu)), silent=silent)
            self._states, self._ctrls, self._gain, self._ks = states, ctrls, Ks, ks
            self._need_recompute = False
            self._step_count = 0
        if self._step_count == self.reuse_feedback:
            self._need_recompute = True
        i = self._step_count
        u = self._ctrls[i] + self._gain[i] @ (state - self._states[i])
        self._step_count += 1
        return u, np.concatenate([state, u])

    def _bare_state(self):
        if self.strict_reference is None:
            return not getattr(self.model, "update_state_reads_state", False)
        return bool(self.strict_reference)

    def traj_to_state(self, traj):
        # The reference returns the bare model state (ilqr.py:96-98), which run() then strips one
        # control too short (:278) -- harmless for models whose update_state ignores the old state
        # (MLP, SINDy, Koopman), a shape error for ARX: see the module docstring.
        state = self.model.traj_to_state(traj)
        if self._bare_state():
            return state
        return np.concatenate([state, traj[-1].ctrl])

    @property
    def state_dim(self):
        return self.model.state_dim + self.system.ctrl_dim

    @staticmethod
    def is_compatible(system, task, model):
        return is_quad_sum(task.get_cost()) and hasattr(model, "stage_into")


class IterativeLQRFactory(ControllerFactory):
    """horizon in [5, 25], default 20 (ilqr.py:36-41)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.Controller = IterativeLQR
        self.name = "IterativeLQR"

    def get_configuration_space(self):
        try:
            from ConfigSpace import ConfigurationSpace
            from ConfigSpace.hyperparameters import UniformIntegerHyperparameter
        except ImportError as e:
            raise ImportError("ConfigSpace is required for get_configuration_space()") from e
        cs = ConfigurationSpace()
        cs.add_hyperparameter(UniformIntegerHyperparameter(name="horizon", lower=5, upper=25,
                                                           default_value=20))
        return cs
