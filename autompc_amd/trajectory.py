"""Discrete-time observation/control history.

Mirrors the reference's ``autompc.Trajectory`` surface that the MPC hot path
touches (reference: autompc/trajectory.py:6-201): ``traj[i].obs``,
``traj[i].ctrl``, ``traj.obs``, ``traj.ctrls``, ``len(traj)``, label indexing
``traj[i, "x"]`` and the module helpers ``zeros`` / ``empty`` / ``extend``.
"""
from collections import namedtuple

import numpy as np

TimeStep = namedtuple("TimeStep", ["obs", "ctrl"])


class Trajectory:
    def __init__(self, system, size, obs, ctrls):
        obs = np.asarray(obs)
        ctrls = np.asarray(ctrls)
        if obs.shape != (size, system.obs_dim):
            raise ValueError("obs is wrong shape")
        if ctrls.shape != (size, system.ctrl_dim):
            raise ValueError("ctrls is wrong shape")
        self._system = system
        self._size = int(size)
        self._obs = obs
        self._ctrls = ctrls

    # -- container protocol -------------------------------------------------
    def __len__(self):
        return self._size

    def _column(self, label):
        sys_ = self._system
        if label in sys_.observations:
            return self._obs, sys_.observations.index(label)
        if label in sys_.controls:
            return self._ctrls, sys_.controls.index(label)
        raise IndexError("Unknown label")

    def _check_time(self, t):
        if not isinstance(t, slice) and not (-self._size <= t < self._size):
            raise IndexError("Time index out of range.")

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            t, label = idx
            self._check_time(t)
            arr, col = self._column(label)
            return arr[t, col]
        if isinstance(idx, slice):
            o, c = self._obs[idx, :], self._ctrls[idx, :]
            return Trajectory(self._system, o.shape[0], o, c)
        self._check_time(idx)
        return TimeStep(self._obs[idx, :], self._ctrls[idx, :])

    def __setitem__(self, idx, val):
        if not isinstance(idx, tuple):
            raise IndexError("Cannot assign to time steps.")
        t, label = idx
        self._check_time(t)
        arr, col = self._column(label)
        arr[t, col] = val

    def __eq__(self, other):
        return (self._system == other.system and self._size == len(other)
                and np.array_equal(self._obs, other.obs)
                and np.array_equal(self._ctrls, other.ctrls))

    # -- accessors ------------------------------------------------------------
    @property
    def system(self):
        return self._system

    @property
    def size(self):
        return self._size

    @property
    def obs(self):
        return self._obs

    @obs.setter
    def obs(self, value):
        if value.shape != self._obs.shape:
            raise ValueError("obs is wrong shape")
        self._obs = value[:]

    @property
    def ctrls(self):
        return self._ctrls

    @ctrls.setter
    def ctrls(self, value):
        if value.shape != self._ctrls.shape:
            raise ValueError("ctrls is wrong shape")
        self._ctrls = value[:]

    def __repr__(self):
        return "Trajectory(len=%d, %r)" % (self._size, self._system)


def zeros(system, size):
    return Trajectory(system, size, np.zeros((size, system.obs_dim)),
                      np.zeros((size, system.ctrl_dim)))


def empty(system, size):
    return Trajectory(system, size, np.empty((size, system.obs_dim)),
                      np.empty((size, system.ctrl_dim)))


def extend(traj, obs, ctrls):
    o = np.concatenate([traj.obs, np.asarray(obs).reshape(-1, traj.system.obs_dim)])
    c = np.concatenate([traj.ctrls, np.asarray(ctrls).reshape(-1, traj.system.ctrl_dim)])
    return Trajectory(traj.system, o.shape[0], o, c)
