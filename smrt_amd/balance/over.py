# -*- coding: utf-8 -*-
"""``over_sample_balance`` (reference: smrt/balance/over.py).  Pure row selection: the draws happen on the caller's
numpy stream exactly as there, but rows are only tracked as indices into ``X`` and the balanced matrix is written
once by the library's multi-threaded gather instead of one ``np.vstack`` of the whole matrix per round."""
from __future__ import absolute_import, division, print_function

import numpy as np

from .. import _hip
from ..utils import DEFAULT_SEED, get_random_state
from .base import validate_X_y_ratio_classes

__all__ = [
    'over_sample_balance'
]


def over_sample_balance(X, y, balance_ratio=0.2, random_state=DEFAULT_SEED):
    """Re-draw rows of every minority class (without replacement per round, rounds repeated as needed) until it has
    ``max(int(balance_ratio * majority_count), 1)`` rows.  Returns ``(X_balanced, y_balanced)``: the originals in
    order, then the re-drawn rows class by class.

    A non-majority class that already has more rows than the target is left alone (the reference's loop waits for the
    shortfall to be exactly zero, over.py:70-74, and never ends for such a class)."""
    random_state = get_random_state(random_state).state                  # over.py:43: before validation

    X, y, n_classes, present_classes, counts, majority_label, target_count = \
        validate_X_y_ratio_classes(X, y, balance_ratio, copy=False)

    y_codes = np.searchsorted(present_classes, y)
    src = np.arange(X.shape[0], dtype=np.int64)            # output row -> row of X
    out_codes = y_codes.copy()
    for code, label in enumerate(present_classes):
        if label == majority_label:
            continue
        while True:
            mask = out_codes == code
            n_req = target_count - int(mask.sum())
            if n_req <= 0:
                break
            idcs = np.arange(src.shape[0])[mask]           # positions in the growing output (copies included)
            picked = random_state.permutation(idcs)[:n_req]
            src = np.concatenate([src, src[picked]])
            out_codes = np.concatenate([out_codes, np.full(picked.shape[0], code, dtype=out_codes.dtype)])
    return _hip.assemble_rows(X, None, src, synth_dtype=np.float64), present_classes[out_codes]
