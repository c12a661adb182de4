"""Input validation shared by the balancers (reference: smrt/balance/base.py:15-64)."""
from __future__ import annotations

import numpy as np
from sklearn.utils import column_or_1d
from sklearn.utils.multiclass import type_of_target
from sklearn.utils.validation import check_array

from ..utils import NPDTYPE, validate_float

__all__ = ["MAX_N_CLASSES", "MIN_N_SAMPLES", "validate_X_y_ratio_classes"]

MAX_N_CLASSES = 100     # most distinct labels a balancer accepts
MIN_N_SAMPLES = 2       # fewest rows any class may have


def validate_X_y_ratio_classes(X, y, ratio, copy=True, keep_float32=False):
    """Check inputs and derive the class bookkeeping every balancer needs.

    Returns ``(X, y, n_classes, present_classes, counts, majority_label, target_count)`` where ``X`` is a
    fresh fp64 copy (``copy=False``: no copy when ``X`` already is a C-contiguous fp64 array; ``keep_float32``: an fp32
    array is validated as it is -- ``smrt_balance`` converts while it writes its output, in one pass), ``present_classes``/``counts`` come from ``np.unique`` (sorted labels), the majority
    is the FIRST class with the largest count and ``target_count = max(int(ratio * majority_count), 1)``.
    Raises ``ValueError`` for a ratio outside (0, 1], a target that is not binary/multiclass, more than
    ``MAX_N_CLASSES`` labels, or a class with fewer than ``MIN_N_SAMPLES`` rows.
    """
    validate_float(ratio, "balance_ratio")
    dtype = [np.float64, np.float32] if (keep_float32 and getattr(X, "dtype", None) == np.float32) else NPDTYPE
    # sklearn's finiteness sweep is one core's work (most of this function at 1M x 64); a C-contiguous float matrix is
    # swept by the library on every core instead, and sklearn is only asked to look when there is something to report
    fast = (isinstance(X, np.ndarray) and X.ndim == 2 and X.flags.c_contiguous and X.size >= (1 << 20)
            and X.dtype in (np.float32, np.float64))
    if fast:
        from .. import _hip
        fast = _hip.all_finite(X)
    X = check_array(X, accept_sparse=False, dtype=dtype, ensure_2d=True, copy=copy, order="C",
                    ensure_all_finite=not fast)
    y = column_or_1d(check_array(y, accept_sparse=False, ensure_2d=False, dtype=None), warn=False)

    small_ints = y.dtype.kind in "iu" and y.size > 0 and int(y.min()) >= 0 and int(y.max()) < (1 << 16)
    if small_ints:
        # non-negative integer labels: one counting pass gives what type_of_target + np.unique (two sorts of y) give --
        # integer targets are 'binary' with up to two distinct values and 'multiclass' beyond (sklearn.utils.multiclass)
        tally = np.bincount(y.astype(np.intp, copy=False))       # (uint64 labels do not cast to intp 'safely'; the range is checked)
        present_classes = np.flatnonzero(tally).astype(y.dtype)
        counts = tally[present_classes]
        kind = "binary" if len(present_classes) <= 2 else "multiclass"
    else:
        kind = type_of_target(y)
    if kind not in ("binary", "multiclass"):
        raise ValueError("balancers only support (binary, multiclass) targets, but got %r" % kind)

    if not small_ints:
        present_classes, counts = np.unique(y, return_counts=True)
    n_classes = len(present_classes)
    if n_classes > MAX_N_CLASSES:
        raise ValueError("balancers currently only support a maximum of %i unique class labels, but %i were "
                         "identified." % (MAX_N_CLASSES, n_classes))

    top = int(np.argmax(counts))
    majority_label, majority_count = present_classes[top], counts[top]
    target_count = max(int(ratio * majority_count), 1)

    if counts.min() < MIN_N_SAMPLES:
        raise ValueError("All label counts must be >= %i" % MIN_N_SAMPLES)
    return X, y, n_classes, present_classes, counts, majority_label, target_count


# the reference spells it with a leading underscore (smrt/balance/base.py:19)
_validate_X_y_ratio_classes = validate_X_y_ratio_classes
