"""MPPI controller whose solve runs on MI355X (HIP rollout + update kernels).

Drop-in for the reference's ``autompc.control.MPPI`` / ``MPPIFactory`` (reference:
autompc/control/mppi.py:26-64 factory, :66-181 controller): same keyword
hyper-parameters (``horizon``, ``num_path``, ``sigma``, ``lmda``, ``seed``,
``niter``), same ``run`` / ``reset`` / ``traj_to_state`` / ``state_dim``, same
state (``act_sequence`` in units of ``umax``, random-initialised).

What runs where
  host   the noise draw, in parity mode: ``np.random.normal(scale=sqrt(sigma),
         size=(num_path, H, nu))`` from numpy's GLOBAL legacy stream -- the same call,
         order and shape as mppi.py:21-24,126 (with the trailing 1 generalised to
         ``nu``: the reference is only correct for ctrl_dim == 1)
  device everything else: warm-start shift, clipping, batched surrogate rollout, stage /
         action / terminal costs, softmin weights, sequence update (ampc_mppi_solve)

``noise="device"`` replaces the host draw with an on-device Philox stream
(statistically equivalent, not bit-identical; no PCIe traffic per solve).
"""
import numpy as np

from .. import _lib, _npstate
from ..costs.blocks import is_mppi_cost, mppi_cost_parts, quad_sum_block
from .controller import Controller, ControllerFactory


def _stage_cost(handle, cost, obs_dim, ctrl_dim, indicators=False):
    """Hand `cost` -- a QuadCost or any (nested) sum of QuadCosts, this package's or the
    reference's own objects, shared goal or not -- to the device as one affine-quadratic block
    (costs/blocks.py; mppi.py:73-82 and ilqr.py:124-129 evaluate it term by term).  Returns the
    block (its ``terminal_goal`` flag matters to iLQR only).
    indicators (MPPI only): the sum may also hold threshold / box terms (thresh_cost.py:8-83), which the
    rollout kernels add to the stage cost (ampc_set_indicator_costs); iLQR needs Hessians they do not have."""
    if indicators:
        blk, terms = mppi_cost_parts(cost, obs_dim, ctrl_dim)
    else:
        blk, terms = quad_sum_block(cost, obs_dim, ctrl_dim), None
    handle.set_cost_blocks(blk["Q"], blk["R"], blk["F"], blk["goal"], blk["lin"], blk["lin_term"],
                           blk["consts"])
    handle.set_indicator_costs(terms)
    return blk


class _ActSequence(np.ndarray):
    """The host copy of a controller's warm-start sequence, handed out by ``MPPI.act_sequence``:
    a plain writable array (the reference exposes a plain attribute that callers edit in place,
    e.g. ``ctl.act_sequence[:] = 0``) that tells its controller when it was written to, so the
    edit reaches the device copy before the next solve.

    A view may be HELD across ``run()`` calls (in the reference the attribute is one array mutated in
    place, so a held reference stays current).  Here the live copy moves to the device; a write
    through a view whose memory is no longer the current sequence first brings that memory up to date
    (one download), then applies -- it never resurrects the pre-``run()`` sequence.  Reads through a
    held view are NOT refreshed: read ``ctl.act_sequence`` again."""

    def __new__(cls, array, owner):
        obj = np.asarray(array).view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)

    @staticmethod
    def _root(a):
        """The array that owns the memory `a` views."""
        while isinstance(getattr(a, "base", None), np.ndarray):
            a = a.base
        return a

    def __setitem__(self, key, value):
        o = self._owner
        if o is not None:
            root = self._root(self)
            if root is not self._root(o._act_host) or not (o._act_dirty or o._act_synced):
                cur = o._current_sequence()
                if root.size != cur.size or not root.flags.c_contiguous:
                    raise RuntimeError("stale act_sequence view: read ctl.act_sequence again")
                mine = root.reshape(cur.shape)           # (a view: the held array's own memory)
                if root is not self._root(cur):
                    mine[...] = cur
                o._act_host, o._act_synced = mine, True
        super().__setitem__(key, value)
        if o is not None:
            o._act_dirty = True

    def __reduce__(self):                      # pickles / deep-copies as a plain array
        return np.asarray(self).copy().__reduce__()


class MPPI(Controller):
    def __init__(self, system, task, model, **kwargs):
        super().__init__(system, task, model)
        if not hasattr(model, "stage_into"):
            raise TypeError("MPPI needs a device-stageable model (autompc_amd.sysid.MLP); "
                            "there is no CPU fallback")
        self.kwargs = kwargs
        self.dim_state, self.dim_ctrl = model.state_dim, system.ctrl_dim
        self.seed = kwargs.get("seed", 0)
        self.H = int(kwargs.get("horizon", 20))
        self.num_path = int(kwargs.get("num_path", 1000))
        self.num_iter = kwargs.get("niter", 1)      # stored, unused -- as in the reference (:92,105)
        self.niter = 1
        self.sigma = kwargs.get("sigma", 1)
        self.lmda = kwargs.get("lmda", 1.0)
        self.noise = kwargs.get("noise", "numpy")
        self.precision = kwargs.get("precision", getattr(model, "precision", "f64"))
        self.device = kwargs.get("device", getattr(model, "device", 0))
        self.per_particle_terminal = bool(kwargs.get("per_particle_terminal", False))
        if self.noise not in ("numpy", "numpy_host", "numpy_device", "device"):
            raise ValueError("noise must be 'numpy' (the reference's draw from numpy's global legacy "
                             "stream, bit for bit; generated on the device when that is provably exact, "
                             "on the host otherwise), 'numpy_host' (always the host draw), "
                             "'numpy_device' (always on the device) or 'device' (Philox, fast)")
        bounds = task.get_ctrl_bounds()
        self.umin = bounds[:, 0].copy()
        self.umax = bounds[:, 1].copy()
        self.ctrl_scale = self.umax
        self._scale = np.sqrt(self.sigma)
        self._handle = None
        self._plan = None
        self._jit_pending = False
        self._init_sequence()

    # -- construction-time state (mppi.py:96-105) ------------------------------------
    def _init_sequence(self):
        self._act_host = np.random.normal(scale=self._scale, size=(self.H, self.dim_ctrl))
        self._act_dirty = True        # host copy is newer than the device copy
        self._act_synced = False      # host copy equals the device copy (set by a download)
        self.cur_step = 0

    def _device(self):
        if self._plan is not None and self._jit_pending:
            # the kernels specialised for this model's shape were still compiling when the plan
            # was made (csrc/jit_host.hpp): switch over once they are ready -- same results, the
            # warm start moves with it
            st = self._handle.jit_status()[0]
            if st == 2:
                self._current_sequence()
                self._plan.close()
                self._plan = None
            self._jit_pending = st == 1
        if self._handle is None:
            h = _lib.Handle(self.device, self.precision)
            self.model.stage_into(h)
            _stage_cost(h, self.task.get_cost(), self.system.obs_dim, self.dim_ctrl, indicators=True)
            h.set_ctrl_bounds(self.umin, self.umax)
            self._handle = h
        if self._plan is None:
            term = _lib.TERM_PER_PARTICLE if self.per_particle_terminal else _lib.TERM_REFERENCE
            self._plan = _lib.MppiPlan(self._handle, [self.num_path], [self.H], [self.sigma], [self.lmda],
                                       term_mode=term)
            self._act_dirty = True
            self._jit_pending = self._plan.kernel_kind() in (0, 3) and self._handle.jit_status()[0] == 1
        return self._plan

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_act_host"] = np.array(self._current_sequence())
        state["_handle"] = state["_plan"] = None
        state["_act_dirty"], state["_act_synced"] = True, False
        return state

    @property
    def act_sequence(self):
        """The warm-start sequence (H, nu) in units of umax -- after a solve the live copy is on the
        device, so reading it downloads it.  The array handed out shares memory with the
        controller's host copy and is writable like the reference's plain attribute
        (mppi.py:97-99): item / slice assignment (``ctl.act_sequence[:] = 0``,
        ``ctl.act_sequence[0] += d``) marks the host copy as newer and it is uploaded before the
        next solve.  Writes that bypass ``__setitem__`` (``np.copyto``, ufunc ``out=``) are not
        seen; assign through the setter for those."""
        return _ActSequence(self._current_sequence(), self)

    def _current_sequence(self):
        """The live sequence as a host array (downloaded when the device copy is the newer one)."""
        if self._plan is not None and not self._act_dirty and not self._act_synced:
            a, _, _, _ = self._plan.download(act_seq=True, u=False)
            self._act_host = a.reshape(self.H, self.dim_ctrl)
            self._act_synced = True
        return self._act_host

    @act_sequence.setter
    def act_sequence(self, value):
        self._act_host = np.array(value, dtype=np.float64).reshape(self.H, self.dim_ctrl)
        self._act_dirty, self._act_synced = True, False

    def reset(self):
        self._init_sequence()

    # -- one control step (mppi.py:154-168) ------------------------------------------
    def run(self, constate, new_obs, return_details=False):
        nu = self.system.ctrl_dim
        x0 = self.model.update_state(constate[:-nu], constate[-nu:], new_obs)
        plan = self._device()
        act = self._act_host if self._act_dirty else None
        mode = self.noise
        if mode == "numpy":
            # np.random.normal(scale, size=(N, H, nu)) from the global legacy generator
            # (mppi.py:16-24, :126).  The device reproduces that draw bit for bit when the
            # library has proven its restatement of the host C library's log() against log()
            # itself (ampc_legacy_log_mode) and the global generator is the MT19937 it models.
            # The generator's state is read and updated where numpy keeps it (_npstate) when that
            # layout is recognised, through get_state() / set_state() otherwise.
            ls = state = None
            if _lib.legacy_log_mode() != 0:
                ls = _npstate.get()
                if ls is None:
                    state = np.random.get_state()
            mode = "numpy_device" if ls is not None or (state is not None and state[0] == "MT19937") else "numpy_host"
        else:
            state = None
            ls = _npstate.get() if mode == "numpy_device" else None
        if return_details:
            if mode == "numpy_host":
                eps = np.random.normal(scale=self._scale, size=(self.num_path, self.H, nu))
                plan.upload(x0=x0, act_seq=act, eps=eps)
            elif mode == "numpy_device":
                plan.upload(x0=x0, act_seq=act)
                self._legacy_draw(plan, ls, state)
            else:
                plan.upload(x0=x0, act_seq=act)
                plan.generate_eps(self.seed, self.cur_step)
            self._act_dirty = False
            plan.solve()
            a, u, costs, eps_out = plan.download(costs=True, eps_out=True)
            self.last_costs = costs
            self.last_eps = eps_out.reshape(self.H, self.num_path, nu)
            self._act_host, self._act_synced = a.reshape(self.H, nu), True
        else:
            # the hot call: everything of this control step in one library call (ampc_mppi_run)
            if mode == "numpy_host":
                plan.upload(eps=np.random.normal(scale=self._scale, size=(self.num_path, self.H, nu)))
                u = plan.run(x0, act)
            elif mode == "numpy_device":
                # the same draw from the same global generator state, made on the device; the host
                # generator is then put into the state the draw would have left it in
                if ls is not None:
                    u = plan.run_legacy_inplace(x0, act, ls)
                else:
                    self._legacy_draw(plan, ls, state)
                    u = plan.run(x0, act)
            else:
                u = plan.run(x0, act, philox=(self.seed, self.cur_step))
            self._act_dirty = self._act_synced = False
        self.cur_step += 1
        ret_action = u[0].copy()
        return ret_action, np.concatenate([x0, ret_action])

    @staticmethod
    def _legacy_draw(plan, ls, state):
        if ls is not None:
            plan.legacy_normal_inplace(ls)
        else:
            np.random.set_state(plan.legacy_normal(state if state is not None else np.random.get_state()))

    def traj_to_state(self, traj):
        return np.concatenate([self.model.traj_to_state(traj), traj[-1].ctrl])

    @property
    def state_dim(self):
        return self.model.state_dim + self.system.ctrl_dim

    @staticmethod
    def is_compatible(system, task, model):
        return is_mppi_cost(task.get_cost()) and hasattr(model, "stage_into")


class MPPIFactory(ControllerFactory):
    """Hyper-parameter ranges of the reference's factory (mppi.py:48-64)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.Controller = MPPI
        self.name = "MPPI"

    def get_configuration_space(self):
        try:
            import ConfigSpace as CS
            import ConfigSpace.hyperparameters as CSH
        except ImportError as e:
            raise ImportError("ConfigSpace is required for get_configuration_space()") from e
        cs = CS.ConfigurationSpace()
        cs.add_hyperparameter(CSH.UniformIntegerHyperparameter("horizon", lower=5, upper=30,
                                                               default_value=20))
        cs.add_hyperparameter(CSH.UniformFloatHyperparameter("sigma", lower=1e-4, upper=2.0,
                                                             default_value=1.0))
        cs.add_hyperparameter(CSH.UniformFloatHyperparameter("lmda", lower=0.1, upper=2.0,
                                                             default_value=1.0))
        cs.add_hyperparameter(CSH.UniformIntegerHyperparameter("num_path", lower=100, upper=1000,
                                                               default_value=200))
        return cs
