"""Golden fixtures for smote_balance / over_sample_balance / under_sample_balance, made by running the REFERENCE's own
Python (smrt/balance/smote.py, over.py, under.py) with the sklearn that is installed here (its kd-tree does the
neighbour search).  Build container only: ``python tests/golden/make_sampler_goldens.py``; only the small data files
written next to this script travel.  Same in-memory shims as make_host_goldens.py (imported for that purpose).

Inputs are continuous random rows, so no two distances tie: with ties (e.g. iris, one decimal digit) the neighbour a
kd-tree returns depends on its traversal order, which nothing in the reference pins.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_host_goldens as shims  # noqa: E402,F401  (installs the shims, puts the reference on sys.path)

from smrt.balance import over_sample_balance, smote_balance, under_sample_balance  # noqa: E402


def blobs(rs, counts, d, labels=None):
    Xs, ys = [], []
    for i, n in enumerate(counts):
        Xs.append(rs.normal(size=d) * 2.0 + rs.normal(size=(n, d)))
        ys.append(np.full(n, i if labels is None else labels[i]))
    X, y = np.vstack(Xs), np.concatenate(ys)
    p = rs.permutation(len(y))
    return X[p], y[p]


def main():
    npz, meta = {}, {}
    rs = np.random.RandomState(2024)

    smote_cases = {
        "sm_shuffled": (blobs(rs, [60, 9, 25], 6), dict(balance_ratio=0.8, n_neighbors=5, random_state=42, shuffle=True)),
        "sm_ordered_k3": (blobs(rs, [40, 3, 17], 5), dict(balance_ratio=1.0, n_neighbors=3, random_state=7, shuffle=False)),
        "sm_strings": (blobs(rs, [30, 8, 12], 4, labels=["neg", "pos", "mid"]),
                       dict(balance_ratio=0.5, n_neighbors=10, random_state=1, shuffle=True)),
        "sm_wide": (blobs(rs, [300, 40], 37), dict(balance_ratio=0.5, n_neighbors=5, random_state=3, shuffle=True)),
        "sm_nothing_to_do": (blobs(rs, [20, 20, 20], 3), dict(balance_ratio=0.2, random_state=42, shuffle=True)),
    }
    for name, ((X, y), kw) in smote_cases.items():
        Xo, yo, models = smote_balance(X, y, return_estimators=True, **kw)
        npz[name + "_X"], npz[name + "_y"] = X, y
        npz[name + "_Xout"], npz[name + "_yout"] = Xo, yo
        meta[name] = dict(kw=kw, fitted=[str(k) for k, v in models.items() if v is not None])

    over_cases = {
        "ov_three": (blobs(rs, [50, 10, 20], 4), dict(balance_ratio=1.0, random_state=42)),
        "ov_half": (blobs(rs, [80, 7], 3), dict(balance_ratio=0.5, random_state=5)),
    }
    for name, ((X, y), kw) in over_cases.items():
        Xo, yo = over_sample_balance(X, y, **kw)
        npz[name + "_X"], npz[name + "_y"], npz[name + "_Xout"], npz[name + "_yout"] = X, y, Xo, yo
        meta[name] = dict(kw=kw)

    under_cases = {
        "un_shuffled": (blobs(rs, [50, 10, 20], 4), dict(balance_ratio=1.0, random_state=42, shuffle=True)),
        "un_ordered": (blobs(rs, [90, 12, 30], 3), dict(balance_ratio=0.5, random_state=9, shuffle=False)),
        "un_equal_top": (blobs(rs, [25, 25, 10], 3), dict(balance_ratio=0.5, random_state=4, shuffle=True)),
    }
    for name, ((X, y), kw) in under_cases.items():
        Xo, yo = under_sample_balance(X, y, **kw)
        npz[name + "_X"], npz[name + "_y"], npz[name + "_Xout"], npz[name + "_yout"] = X, y, Xo, yo
        meta[name] = dict(kw=kw)

    X, y = blobs(rs, [30, 10], 3)
    meta["errors"] = {
        "smote_n_neighbors_zero": shims.exc_name(smote_balance, X, y, n_neighbors=0),
        "smote_bad_strategy": shims.exc_name(smote_balance, X, y, strategy="bad-input"),
        "smote_ratio_zero": shims.exc_name(smote_balance, X, y, balance_ratio=0.0),
        "over_ratio_gt_one": shims.exc_name(over_sample_balance, X, y, balance_ratio=1.5),
        "under_single_sample_class": shims.exc_name(under_sample_balance, X, np.array([0] * 39 + [1])),
    }
    np.savez_compressed(os.path.join(HERE, "sampler_goldens.npz"), **npz)
    with open(os.path.join(HERE, "sampler_goldens.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(npz), "arrays;", meta["errors"])


if __name__ == "__main__":
    main()
