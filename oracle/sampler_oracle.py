"""CPU restatement of the reference's other balancers -- TEST INFRASTRUCTURE, never imported by the product path.

    smote_balance          smrt/balance/smote.py:22-98 (per class), :101-260 (driver)
    over_sample_balance    smrt/balance/over.py:20-86
    under_sample_balance   smrt/balance/under.py:18-78

Pinned by tests/golden/sampler_goldens.npz, which the reference itself produced (tests/golden/make_sampler_goldens.py,
sklearn kd-tree neighbour search).  The neighbour search here is a brute-force sweep over the same fp64 sums
``sum_j (a_j - b_j)^2`` accumulated in feature order; equal sums go to the lower row number (a stable sort), which is
a choice of this restatement: the reference leaves ties to the kd-tree's traversal order ("ties unpinned").

Loops are kept per consideration / per round like the reference's, so the numpy random stream is consumed call for
call the way it is there.
"""
import numpy as np


def validate(X, y, ratio):
    """balance/base.py:19-64, the parts these balancers need (the full restatement is pinned in test_host_goldens)."""
    if not (isinstance(ratio, (int, float)) and 0 < ratio <= 1.0):
        raise ValueError("ratio must be a float in (0, 1]")
    X = np.array(X, dtype=np.float64, copy=True)
    y = np.asarray(y)
    if y.ndim == 2 and y.shape[1] == 1:
        y = y.ravel()
    present, counts = np.unique(y, return_counts=True)
    if len(present) > 100:
        raise ValueError("too many classes")
    if np.any(counts < 2):
        raise ValueError("each class needs at least two rows")
    majority = present[np.argmax(counts)]
    target = max(int(ratio * counts.max()), 1)
    return X, y, present, counts, majority, target


def exact_knn(X_sub, k):
    """Indices [n, k] of the k nearest rows (self included), squared distances summed in feature order."""
    n, d = X_sub.shape
    acc = np.zeros((n, n), dtype=np.float64)
    for j in range(d):                                     # sequential accumulation, one rounding per operation
        diff = X_sub[:, j][:, None] - X_sub[:, j][None, :]
        acc = acc + diff * diff
    return np.argsort(acc, axis=1, kind="stable")[:, :k]


def smote_class(X, X_sub, target_count, random_state, strategy, n_neighbors):
    """smote.py:39-98 for one class: returns the synthetic rows appended for it, in order."""
    count = X_sub.shape[0]
    amt_required = target_count - count
    k_neighbors = min(count, n_neighbors + 1)                                   # smote.py:68
    nearest = exact_knn(X_sub, k_neighbors)
    indices = np.arange(count)
    made = []
    while amt_required > 0:
        draw_count = max(1, int(round(amt_required / n_neighbors)))             # smote.py:79
        random_indices = random_state.permutation(indices)[:draw_count]
        blocks = []
        for c in random_indices:
            nbrs = X_sub[nearest[c][1:], :]
            if strategy == "perturb":                                           # smote.py:22-23
                blocks.append((X_sub[c, :] - nbrs) * random_state.rand(nbrs.shape[0], 1) + X_sub[c, :])
            else:                                                               # smote.py:26-29 (midpoint)
                blocks.append((X_sub[c, :] + nbrs) / 2.0)
        synthetic = random_state.permutation(np.vstack(blocks))                 # smote.py:84
        made.append(synthetic[:amt_required])
        amt_required -= synthetic.shape[0]
    return np.vstack(made)


def smote_balance(X, y, balance_ratio=0.2, strategy="perturb", n_neighbors=5, random_state=42, shuffle=True):
    X, y, present, counts, majority, target = validate(X, y, balance_ratio)
    if n_neighbors < 1:
        raise ValueError("n_neighbors must be at least 1")
    if strategy not in ("perturb", "interpolate"):
        raise ValueError("bad strategy")
    rs = np.random.RandomState(random_state)
    codes = np.searchsorted(present, y)
    out_X, out_codes = [X], [codes]
    for code, label in enumerate(present):
        if label == majority or counts[code] >= target:
            continue
        X_sub = X[codes == code, :]
        rows = smote_class(X, X_sub, target, rs, strategy, n_neighbors)
        out_X.append(rows)
        out_codes.append(np.full(rows.shape[0], code, dtype=codes.dtype))
    Xo, yo = np.vstack(out_X), present[np.concatenate(out_codes)]
    order = np.arange(Xo.shape[0])
    if shuffle:
        order = rs.permutation(order)
    return Xo[order, :], yo[order]


def over_sample_balance(X, y, balance_ratio=0.2, random_state=42):
    rs = np.random.RandomState(random_state)                                    # over.py:43 (before validation)
    X, y, present, counts, majority, target = validate(X, y, balance_ratio)
    out_X, out_y = X.copy(), np.searchsorted(present, y)
    for code, label in enumerate(present):
        if label == majority:
            continue
        while True:                                                             # over.py:65-84
            mask = out_y == code
            n_req = target - mask.sum()
            if n_req <= 0:         # the reference tests `== 0` and never terminates for a class above target
                break
            idcs = np.arange(out_X.shape[0])[mask]
            sample = out_X[rs.permutation(idcs), :][:n_req]
            out_X = np.vstack([out_X, sample])
            out_y = np.concatenate([out_y, np.full(sample.shape[0], code, dtype=out_y.dtype)])
    return out_X, present[out_y]


def under_sample_balance(X, y, balance_ratio=0.2, random_state=42, shuffle=True):
    rs = np.random.RandomState(random_state)
    X, y, present, counts, majority, _ = validate(X, y, balance_ratio)

    def reorder(X, y):
        order = np.arange(X.shape[0])
        if shuffle:
            order = rs.permutation(order)
        return X[order, :], y[order]

    sorted_counts = np.sort(counts)
    if sorted_counts[-1] == sorted_counts[-2]:                                  # under.py:63-64
        return reorder(X, y)
    target = max(int(sorted_counts[-2] / balance_ratio), 1)
    idcs = np.arange(X.shape[0])
    mask = y == majority
    remove = rs.permutation(idcs[mask])[:mask.sum() - target]                   # under.py:71
    return reorder(np.delete(X, remove, axis=0), np.delete(y, remove))
