"""Minimal stand-ins for ``gym.spaces`` (gym is not a dependency).  Downstream code of the
reference only dispatches on ``space.__class__.__name__`` and reads ``.shape`` / ``.n``
(reference onpolicy/utils/util.py:32-53, onpolicy/runner/shared/graph_mpe_runner.py:420-433)."""
import numpy as np


class Box(object):
    def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return 'Box%s' % (self.shape,)


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()

    def __repr__(self):
        return 'Discrete(%d)' % self.n
