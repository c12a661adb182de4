# -*- coding: utf-8 -*-
"""``smote_balance`` on the MI355X (reference: smrt/balance/smote.py).

What runs where
  * neighbour search (smote.py:53-69, sklearn ``NearestNeighbors`` on a kd-tree there): an exact fp64 brute-force
    sweep on the GPU (``smrt_knn_f64``) -- the same ``sum_j (a_j - b_j)^2`` sums the tree metric accumulates, so the
    same neighbours in the same order wherever no two distances are exactly equal (ties go to the lower row here; the
    reference leaves them to the tree's traversal order);
  * the synthetic rows (smote.py:22-29,84-89): one ``smrt_smote_rows_f64`` launch per class, rows already in their
    final (shuffled, truncated) order;
  * every random draw: the host, on the caller's numpy stream, call for call in the reference's order (selection
    permutation, one ``rand`` per neighbour, row shuffle -- per round; output permutation last);
  * output: ``[X ; synthetic][order]`` written once by ``smrt_assemble_rows_f64``.

There is no CPU fallback: without the HIP library or a GPU this raises.  Only the Euclidean metric is provided
(``metric`` 'minkowski' with ``p=2``, 'euclidean' or 'l2'); ``algorithm``, ``leaf_size`` and ``n_jobs`` select how
sklearn searches, not what it finds, and are accepted and ignored.
"""
from __future__ import absolute_import, division, print_function

import numpy as np

from .. import _hip
from ..utils import DEFAULT_SEED, get_random_state
from .base import validate_X_y_ratio_classes

__all__ = [
    'smote_balance'
]

STRATEGIES = ('perturb', 'interpolate')
MAX_NEIGHBORS = 63                      # smrt_knn_f64 keeps at most 64 rows per query (the row itself included)


class HipNearestNeighbors(object):
    """What ``return_estimators=True`` hands back per class: the fitted rows (device-resident fp64) and a
    ``kneighbors`` over them -- the two things a caller can do with the reference's fitted ``NearestNeighbors``."""

    def __init__(self, n_neighbors=5, device=0):
        self.n_neighbors = int(n_neighbors)
        self.device = int(device)
        self._fit_X = None

    def fit(self, X):
        import torch
        self._fit_X = _hip._dev_f64(X, torch.device("cuda", self.device))
        self.n_samples_fit_ = int(self._fit_X.shape[0])
        return self

    def kneighbors(self, X=None, n_neighbors=None, return_distance=True):
        if self._fit_X is None:
            raise ValueError("this HipNearestNeighbors instance is not fitted yet")
        k = self.n_neighbors if n_neighbors is None else int(n_neighbors)
        if X is not None:
            return _hip.knn_f64(self._fit_X, k, Q_rows=X, device=self.device, return_distance=return_distance)
        # sklearn semantics for X=None: neighbours of the fitted rows, each row itself excluded
        if k + 1 > self.n_samples_fit_:
            raise ValueError("Expected n_neighbors < n_samples_fit")
        got = _hip.knn_f64(self._fit_X, k + 1, device=self.device, return_distance=return_distance)
        dist, idx = got if return_distance else (None, got)
        own = idx == np.arange(idx.shape[0])[:, None]
        own[~own.any(axis=1), -1] = True                  # an exact duplicate displaced the row itself: drop the farthest
        keep = ~own
        idx = idx[keep].reshape(idx.shape[0], k)
        return (dist[keep].reshape(idx.shape[0], k), idx) if return_distance else idx


def _check_metric(metric, p, metric_params):
    if callable(metric) or metric_params:
        raise NotImplementedError("only the Euclidean metric is provided on the HIP backend")
    if not ((metric == 'minkowski' and p == 2) or metric in ('euclidean', 'l2')):
        raise NotImplementedError("metric=%r, p=%r: only the Euclidean metric is provided on the HIP backend" % (metric, p))


def _plan_class(n_sub, nearest, amt_required, n_neighbors, random_state, perturb):
    """The random draws of smote.py:76-96 for one class, vectorised per round: returns (base rows, neighbour rows,
    factors) of the synthetic rows in their final order."""
    indices = np.arange(n_sub)
    kn = nearest.shape[1] - 1
    bases, nbrs, facs = [], [], []
    while amt_required > 0:
        draw_count = max(1, int(round(amt_required / n_neighbors)))
        chosen = random_state.permutation(indices)[:draw_count]
        base = np.repeat(chosen, kn)
        nbr = nearest[chosen, 1:].ravel()
        # one rand(kn, 1) per consideration, back to back == one draw of draw_count * kn doubles
        fac = random_state.random_sample(chosen.shape[0] * kn) if perturb else None
        order = random_state.permutation(base.shape[0])[:amt_required]        # the row shuffle, then the truncation
        bases.append(base[order])
        nbrs.append(nbr[order])
        if perturb:
            facs.append(fac[order])
        amt_required -= base.shape[0]
    return np.concatenate(bases), np.concatenate(nbrs), (np.concatenate(facs) if perturb else None)


def smote_balance(X, y, return_estimators=False, balance_ratio=0.2, strategy='perturb', n_neighbors=5,
                  algorithm='kd_tree', leaf_size=30, p=2, metric='minkowski', metric_params=None, n_jobs=1,
                  random_state=DEFAULT_SEED, shuffle=True):
    """Synthetic Minority Over-sampling: every class below ``max(int(balance_ratio * majority_count), 1)`` rows is
    topped up with rows built from a randomly chosen row and each of its ``n_neighbors`` nearest class-mates --
    ``(x - neighbour) * u + x`` with ``u ~ U[0, 1)`` ('perturb'), or their midpoint ('interpolate').

    Returns ``(X_balanced, y_balanced)`` (fp64 rows; originals first unless ``shuffle``), plus
    ``{label: fitted neighbour model or None}`` when ``return_estimators``.
    """
    X, y, n_classes, present_classes, counts, majority_label, target_count = \
        validate_X_y_ratio_classes(X, y, balance_ratio, copy=False)

    if n_neighbors < 1:
        raise ValueError('n_neighbors must be at least 1')
    if strategy not in STRATEGIES:
        raise ValueError('strategy must be one of %r' % (STRATEGIES,))
    _check_metric(metric, p, metric_params)
    if n_neighbors > MAX_NEIGHBORS:
        raise NotImplementedError("n_neighbors > %d is not provided on the HIP backend" % MAX_NEIGHBORS)

    random_state = get_random_state(random_state).state
    y_codes = np.searchsorted(present_classes, y)          # == LabelEncoder().fit(present_classes).transform(y)

    import torch
    device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    models, pieces, piece_codes = dict(), [], []
    for code, label in enumerate(present_classes):
        models[label] = None
        count = int(counts[code])
        if label == majority_label or count >= target_count:
            continue
        X_sub = X[y_codes == code, :]
        model = HipNearestNeighbors(n_neighbors=n_neighbors, device=device).fit(X_sub)
        k_neighbors = min(count, n_neighbors + 1)           # the row itself comes back first (smote.py:59-69)
        nearest = model.kneighbors(model._fit_X, n_neighbors=k_neighbors, return_distance=False)
        base, nbr, fac = _plan_class(count, nearest, target_count - count, n_neighbors, random_state,
                                     strategy == 'perturb')
        pieces.append(_hip.smote_rows(model._fit_X, base, nbr, fac, mode=strategy))
        piece_codes.append(np.full(base.shape[0], code, dtype=y_codes.dtype))
        models[label] = model

    synth = np.vstack(pieces) if pieces else None
    y_out = present_classes[np.concatenate([y_codes] + piece_codes)]
    n_total = y_out.shape[0]
    output_order = _hip.rng_permutation(random_state, n_total) if shuffle else None          # == permutation(np.arange(n_total))
    X_out = _hip.assemble_rows(X, synth, output_order, synth_dtype=np.float64)
    if shuffle:
        y_out = y_out[output_order]
    if return_estimators:
        return X_out, y_out, models
    return X_out, y_out
