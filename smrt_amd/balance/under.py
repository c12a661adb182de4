# -*- coding: utf-8 -*-
"""``under_sample_balance`` (reference: smrt/balance/under.py).  Row removal and the optional shuffle are composed
into one index vector; the library's multi-threaded gather writes the result once."""
from __future__ import absolute_import, division, print_function

import numpy as np

from .. import _hip
from ..utils import DEFAULT_SEED, get_random_state
from .base import validate_X_y_ratio_classes

__all__ = [
    'under_sample_balance'
]


def under_sample_balance(X, y, balance_ratio=0.2, random_state=DEFAULT_SEED, shuffle=True):
    """Drop randomly chosen rows of the majority class until the second most populous class stands at
    ``balance_ratio`` to it (``target = max(int(second_count / balance_ratio), 1)`` majority rows stay).  When the two
    largest classes are equally large nothing is dropped.  Output rows keep their order unless ``shuffle``."""
    random_state = get_random_state(random_state).state

    X, y, n_classes, present_classes, counts, majority_label, _ = \
        validate_X_y_ratio_classes(X, y, balance_ratio, copy=False)

    keep = np.arange(X.shape[0], dtype=np.int64)
    sorted_counts = np.sort(counts)
    if sorted_counts[-1] != sorted_counts[-2]:                           # under.py:63-64
        target_count = max(int(sorted_counts[-2] / balance_ratio), 1)
        mask = y == majority_label
        n_major = int(mask.sum())
        remove = random_state.permutation(keep[mask])[:n_major - target_count]      # under.py:71, slice semantics kept
        keep = np.delete(keep, remove)
    if shuffle:
        keep = keep[random_state.permutation(np.arange(keep.shape[0]))]
    return _hip.assemble_rows(X, None, keep, synth_dtype=np.float64), y[keep]
