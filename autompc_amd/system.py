"""Robot system descriptor: labelled observation / control dimensions and dt.

Duck-type compatible with the reference's ``autompc.System``
(reference: autompc/system.py:3-79): ``obs_dim``, ``ctrl_dim``,
``observations``, ``controls``, ``dt``.  Controllers and models in this package
only ever read those five attributes, so a reference ``System`` object can be
passed wherever one of these is expected and vice versa.
"""


class System:
    def __init__(self, observations, controls, dt=None):
        obs = list(observations)
        ctl = list(controls)
        labels = obs + ctl
        if len(set(labels)) != len(labels):
            raise ValueError("Observation and control labels must be unique")
        self._obs_labels = tuple(obs)
        self._ctrl_labels = tuple(ctl)
        self.dt = dt

    @property
    def observations(self):
        return list(self._obs_labels)

    @property
    def controls(self):
        return list(self._ctrl_labels)

    @property
    def obs_dim(self):
        return len(self._obs_labels)

    @property
    def ctrl_dim(self):
        return len(self._ctrl_labels)

    def __eq__(self, other):
        return (list(getattr(other, "observations", ())) == self.observations
                and list(getattr(other, "controls", ())) == self.controls)

    def __hash__(self):
        return hash((self._obs_labels, self._ctrl_labels))

    def __repr__(self):
        return "System(obs=%d, ctrl=%d, dt=%r)" % (self.obs_dim, self.ctrl_dim, self.dt)
