"""Host utilities shared by the estimator and the balancer.

Mirrors the reference's ``smrt/utils.py`` interface (names, argument meaning, errors):
``SeededRandomState`` (utils.py:30-57), ``get_random_state`` (:60-70),
``validate_float`` (:73-77), constants (:20-24).
"""
from __future__ import annotations

import numpy as np
from numpy.random import RandomState

__all__ = ["SeededRandomState", "get_random_state", "validate_float", "overrides",
           "NPDTYPE", "DEVICE_DTYPE", "DEFAULT_L2", "DEFAULT_DROPOUT", "DEFAULT_SEED"]

NPDTYPE = np.float64          # host dtype            (reference utils.py:21)
DEVICE_DTYPE = np.float32     # compute dtype, fp32   (reference utils.py:20, DTYPE = tf.float32)
DEFAULT_L2 = 0.0001
DEFAULT_DROPOUT = 1.0
DEFAULT_SEED = 42


class SeededRandomState(object):
    """A ``numpy.random.RandomState`` that remembers the seed that built it.

    ``seed=None`` means ``DEFAULT_SEED``.  A ``RandomState`` instance is accepted and
    reduced to its first Mersenne-Twister word, as the reference intends
    (utils.py:46-47; there the extracted ``numpy.uint32`` then trips the int check --
    here numpy integers are accepted, a documented divergence).
    """

    def __init__(self, seed=DEFAULT_SEED):
        if seed is None:
            seed = DEFAULT_SEED
        if isinstance(seed, RandomState):
            seed = int(seed.get_state()[1][0])
        if isinstance(seed, (bool, np.bool_)) or not isinstance(seed, (int, np.integer)):
            raise TypeError("seed must be an int, but got type=%s" % type(seed))
        self.seed_value = int(seed)
        self.rs = RandomState(self.seed_value)

    @property
    def state(self):
        return self.rs


def get_random_state(seed):
    """Pass a ``SeededRandomState`` through, otherwise build one (utils.py:60-70)."""
    if isinstance(seed, SeededRandomState):
        return seed
    return SeededRandomState(seed)


def validate_float(ratio, name, upper_bound=1.0, ltet=True):
    """``0 < ratio <= upper_bound`` (``ltet``) or ``0 < ratio < upper_bound``; ValueError otherwise."""
    ok = (0.0 < ratio <= upper_bound) if ltet else (0.0 < ratio < upper_bound)
    if not ok:
        raise ValueError("Expected 0 < %s %s %r, but got %r" % (name, "<=" if ltet else "<", upper_bound, ratio))


def overrides(interface_class):
    """Decorator asserting that the decorated method exists on ``interface_class`` (utils.py:80-124)."""
    def check(method):
        assert method.__name__ in dir(interface_class), \
            "%s.%s must override a super method!" % (interface_class.__name__, method.__name__)
        return method
    return check
