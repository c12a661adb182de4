"""numpy restatement of the reference's smrt_balance host logic.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Pinned against fixtures produced
by the reference's own Python (tests/golden/make_host_goldens.py -> host_goldens.*).

Citations are relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np
from sklearn.utils.multiclass import type_of_target

MAX_N_CLASSES = 100      # smrt/balance/base.py:15
MIN_N_SAMPLES = 2        # smrt/balance/base.py:16


def validate_X_y_ratio(X, y, ratio):
    """smrt/balance/base.py:19-64."""
    if not (0.0 < ratio <= 1.0):                                   # utils.py:73-77 (ltet=True)
        raise ValueError("Expected 0 < balance_ratio <= 1.0, but got %r" % ratio)
    X = np.array(X, dtype=np.float64, copy=True)                   # check_array(copy=True), base.py:24
    if X.ndim != 2:
        raise ValueError("Expected 2D array")
    if not np.all(np.isfinite(X)):
        raise ValueError("Input contains NaN or infinity")
    y = np.asarray(y)
    if y.ndim == 2 and y.shape[1] == 1:
        y = y.ravel()                                              # column_or_1d, base.py:26
    if y.ndim != 1:
        raise ValueError("y should be a 1d array")
    y_type = type_of_target(y)
    if y_type not in ("multiclass", "binary"):                     # base.py:30-34
        raise ValueError("balancers only support (binary, multiclass), but got %r" % y_type)
    present, counts = np.unique(y, return_counts=True)             # base.py:36
    if len(present) > MAX_N_CLASSES:                               # base.py:40-42
        raise ValueError("too many classes")
    maj_idx = int(np.argmax(counts))                               # first max wins, base.py:45
    target = max(int(ratio * counts[maj_idx]), 1)                  # base.py:47
    if any(c < MIN_N_SAMPLES for c in counts):                     # base.py:61-62
        raise ValueError("All label counts must be >= %i" % MIN_N_SAMPLES)
    return X, y, len(present), present, counts, present[maj_idx], target


def balance(X, y, make_estimator, balance_ratio=0.2, random_state=42, shuffle=True,
            gen_from_samples=False, generate_args=None, prefit_estimators=None, return_estimators=False):
    """smrt/balance/smrt.py:170-250.

    ``make_estimator(X_sub, seeded_state)`` stands for
    ``VariationalAutoEncoder(..., random_state=seeded_random_state).fit(X_sub)`` (smrt.py:212-220);
    ``random_state`` here is the ONE numpy RandomState shared by everything (smrt.py:174-175).

    Divergence kept on purpose: smrt.py:207 indexes the original X with the label vector that
    smrt.py:238 has already extended, which only works under numpy<1.13 (extra entries are all
    False there); the mask is taken over the original labels, which is what that computed.
    """
    X, y, _, present, counts, majority, target = validate_X_y_ratio(X, y, balance_ratio)
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(
        42 if random_state is None else random_state)
    y_idx = np.searchsorted(present, y)                            # LabelEncoder on sorted uniques
    n0 = X.shape[0]
    generate_args = {} if generate_args is None else generate_args
    if prefit_estimators is None:
        prefit_estimators = {}
    elif not isinstance(prefit_estimators, dict):
        raise ValueError("if prefit estimators are provided, it must be a dict")
    encoders, blocks, labels = {}, [X], [y_idx]
    for i, label in enumerate(present):
        if label == majority or counts[i] >= target:               # smrt.py:200-203
            encoders[label] = None
            continue
        X_sub = X[y_idx[:n0] == i, :]
        est = prefit_estimators.get(label, None)
        if est is None:
            est = make_estimator(X_sub, rs)
        encoders[label] = est
        need = target - X_sub.shape[0]
        if not gen_from_samples:
            synth = est.generate(n=need, **generate_args)
        else:
            pick = rs.choice(np.arange(X_sub.shape[0]), size=need, replace=True)
            synth = est.generate_from_sample(X_sub[pick, :], **generate_args)
        blocks.append(np.asarray(synth))
        labels.append(np.ones(need, dtype=np.int16) * i)
    X_out = np.vstack(blocks)                                       # fp64 <- fp32 rows
    y_out = present[np.concatenate(labels)]
    order = np.arange(X_out.shape[0])
    if shuffle:
        order = rs.permutation(order)
    if return_estimators:
        return X_out[order, :], y_out[order], encoders
    return X_out[order, :], y_out[order]
