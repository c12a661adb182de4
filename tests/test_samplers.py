"""smote_balance / over_sample_balance / under_sample_balance.

CPU: the oracle restatement against fixtures the reference itself produced (tests/golden/make_sampler_goldens.py),
and the two pure row-selection balancers of the product (they need the library's host gather, not a GPU).
GPU: the neighbour search and SMOTE rows through the C ABI, bit for bit against the same fixtures and the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import sampler_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "sampler_goldens.npz"), allow_pickle=False)
with open(os.path.join(HERE, "golden", "sampler_goldens.json")) as f:
    META = json.load(f)
CASES = sorted(k for k in META if k != "errors")


def _case(name):
    return G[name + "_X"], G[name + "_y"], G[name + "_Xout"], G[name + "_yout"], META[name]["kw"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_fixtures_bit_for_bit(name):
    X, y, Xo, yo, kw = _case(name)
    fn = {"sm": so.smote_balance, "ov": so.over_sample_balance, "un": so.under_sample_balance}[name[:2]]
    Xb, yb = fn(X, y, **kw)
    assert Xb.dtype == np.float64 and np.array_equal(Xb, Xo) and np.array_equal(yb, yo)


@pytest.mark.parametrize("name", [c for c in CASES if c[:2] in ("ov", "un")])
def test_over_and_under_match_reference_fixtures(name):
    from smrt_amd.balance import over_sample_balance, under_sample_balance
    X, y, Xo, yo, kw = _case(name)
    fn = over_sample_balance if name.startswith("ov") else under_sample_balance
    X_in = X.copy()
    Xb, yb = fn(X, y, **kw)
    assert Xb.dtype == np.float64 and np.array_equal(Xb, Xo) and np.array_equal(yb, yo)
    assert np.array_equal(X, X_in)                                   # the caller's matrix is not touched


def test_over_under_reference_test_shapes():
    """test_over.py:14-27 and test_under.py:13-35 on the iris subset they use (50 / 10 / 20 rows)."""
    from sklearn.datasets import load_iris
    from smrt_amd.balance import over_sample_balance, under_sample_balance
    iris = load_iris()
    idx = list(range(50)) + list(range(50, 60)) + list(range(100, 120))
    X, y = iris.data[idx, :], iris.target[idx]
    Xb, yb = over_sample_balance(X, y, balance_ratio=1.0, random_state=42)
    assert Xb.shape[0] == 150 and all(c == 50 for c in np.unique(yb, return_counts=True)[1])
    np.testing.assert_array_equal(Xb[:80], X)
    Xu, yu = under_sample_balance(X, y, balance_ratio=1.0, random_state=42, shuffle=False)
    labels, counts = np.unique(yu, return_counts=True)
    assert Xu.shape[0] == 50 and dict(zip(labels, counts)) == {0: 20, 1: 10, 2: 20}
    Xs, ys = under_sample_balance(iris.data, iris.target, balance_ratio=1.0, random_state=42, shuffle=False)
    np.testing.assert_array_equal(Xs, iris.data)                      # equal class sizes: nothing is dropped


def test_over_leaves_a_class_above_target_alone():
    """The reference's loop never ends for a minority class above the target (over.py:70-74); here it is skipped."""
    from smrt_amd.balance import over_sample_balance
    rs = np.random.RandomState(0)
    X = rs.rand(100, 3)
    y = np.array([0] * 60 + [1] * 30 + [2] * 10)
    Xb, yb = over_sample_balance(X, y, balance_ratio=0.3, random_state=1)       # target 18: class 1 has 30 already
    assert dict(zip(*np.unique(yb, return_counts=True))) == {0: 60, 1: 30, 2: 18}
    Xo, yo = so.over_sample_balance(X, y, balance_ratio=0.3, random_state=1)
    assert np.array_equal(Xb, Xo) and np.array_equal(yb, yo)


def test_sampler_argument_errors_match_reference():
    from smrt_amd.balance import over_sample_balance, smote_balance, under_sample_balance
    rs = np.random.RandomState(0)
    X, y = rs.rand(40, 3), np.array([0] * 30 + [1] * 10)
    want = META["errors"]
    calls = {
        "smote_n_neighbors_zero": lambda: smote_balance(X, y, n_neighbors=0),
        "smote_bad_strategy": lambda: smote_balance(X, y, strategy="bad-input"),
        "smote_ratio_zero": lambda: smote_balance(X, y, balance_ratio=0.0),
        "over_ratio_gt_one": lambda: over_sample_balance(X, y, balance_ratio=1.5),
        "under_single_sample_class": lambda: under_sample_balance(X, np.array([0] * 39 + [1])),
    }
    for key, call in calls.items():
        with pytest.raises(Exception) as info:
            call()
        assert type(info.value).__name__ == want[key], key
    with pytest.raises(NotImplementedError):
        smote_balance(X, y, metric="manhattan")
    with pytest.raises(NotImplementedError):
        smote_balance(X, y, p=1)


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("sm")])
def test_smote_matches_reference_fixtures_bit_for_bit(name):
    from smrt_amd.balance import smote_balance
    X, y, Xo, yo, kw = _case(name)
    Xb, yb, models = smote_balance(X, y, return_estimators=True, **kw)
    assert Xb.dtype == np.float64 and np.array_equal(yb, yo)
    assert np.array_equal(Xb, Xo)
    assert sorted(str(k) for k, v in models.items() if v is not None) == sorted(META[name]["fitted"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k", [(1, 1, 1), (63, 5, 6), (64, 16, 11), (65, 17, 64), (500, 784, 11), (1777, 64, 6)])
def test_knn_equals_oracle_search(n, d, k):
    """Sizes around the 64-row / 16-feature tiles, the largest k, and the two networks' widths."""
    from smrt_amd import _hip
    rs = np.random.RandomState(n + d)
    X = rs.normal(size=(n, d))
    k = min(k, n)
    dist, idx = _hip.knn_f64(X, k, return_distance=True)
    np.testing.assert_array_equal(idx, so.exact_knn(X, k))
    ref = np.sqrt(((X[:, None, :] - X[idx]) ** 2).sum(-1))
    np.testing.assert_allclose(dist, ref, rtol=1e-12, atol=1e-14)
    assert np.array_equal(idx[:, 0], np.arange(n))                   # every row is its own nearest neighbour


@pytest.mark.gpu
def test_knn_ties_go_to_the_lower_row_and_queries_differ_from_fitted_rows():
    from sklearn.neighbors import NearestNeighbors
    from smrt_amd import _hip
    from smrt_amd.balance.smote import HipNearestNeighbors
    X = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [-1.0, 0.0], [0.0, -1.0], [0.0, 0.0]])
    idx = _hip.knn_f64(X, 6)
    assert idx[0].tolist() == [0, 5, 1, 2, 3, 4] and idx[5].tolist() == [0, 5, 1, 2, 3, 4]
    rs = np.random.RandomState(3)
    F, Q = rs.normal(size=(300, 9)), rs.normal(size=(70, 9))
    model = HipNearestNeighbors(n_neighbors=4).fit(F)
    sk = NearestNeighbors(n_neighbors=4, algorithm="kd_tree").fit(F)
    d_h, i_h = model.kneighbors(Q)
    d_s, i_s = sk.kneighbors(Q)
    np.testing.assert_array_equal(i_h, i_s)
    np.testing.assert_allclose(d_h, d_s, rtol=1e-12)
    np.testing.assert_array_equal(model.kneighbors(return_distance=False), sk.kneighbors(return_distance=False))
    with pytest.raises(_hip.HipLibraryError):
        _hip.knn_f64(F, 65)
    with pytest.raises(_hip.HipLibraryError):
        _hip.knn_f64(X, 7)


@pytest.mark.gpu
def test_smote_reference_test_shapes_and_geometry():
    """test_smote.py:16-44 (counts on the iris subset and the MNIST-shaped set) and test_corner_cases.py:58-70, plus
    what a 'perturb' row must be: x + u (x - neighbour) with u in [0, 1), i.e. beyond x on the line from a class-mate."""
    from sklearn.datasets import load_iris
    from smrt_amd.balance import smote_balance
    iris = load_iris()
    idx = list(range(50)) + list(range(50, 60)) + list(range(100, 120))
    X, y = iris.data[idx, :], iris.target[idx]
    Xb, yb = smote_balance(X, y, balance_ratio=1.0, random_state=42, shuffle=False)
    assert Xb.shape[0] == 150 and all(c == 50 for c in np.unique(yb, return_counts=True)[1])
    np.testing.assert_array_equal(Xb[:80], X)
    Xs, ys = smote_balance(iris.data, iris.target)
    assert Xs.shape[0] == ys.shape[0] == 150
    for shuffle in (True, False):
        for model in (True, False):
            out = smote_balance(iris.data, iris.target, shuffle=shuffle, return_estimators=model)
            assert len(out) == (3 if model else 2)

    rs = np.random.RandomState(2)
    n_major, n_minor, D = 5444, 500, 784
    proto, mask = rs.uniform(size=(2, D)), rs.uniform(size=(2, D)) < 0.2
    Xm = np.vstack([np.clip(proto[c] + 0.15 * rs.normal(size=(n, D)), 0, 1) * mask[c]
                    for c, n in ((0, n_major), (1, n_minor))])
    ym = np.array([0] * n_major + [1] * n_minor)
    Xo, yo = smote_balance(Xm, ym, random_state=42, n_neighbors=10, shuffle=False, balance_ratio=0.5)
    assert Xo.shape[0] == yo.shape[0] == 8166 and (yo == 1).sum() == 2722 and (yo == 0).sum() == 5444
    np.testing.assert_array_equal(Xo[:5944], Xm)
    Xr, yr = so.smote_balance(Xm, ym, random_state=42, n_neighbors=10, shuffle=False, balance_ratio=0.5)
    assert np.array_equal(Xo, Xr) and np.array_equal(yo, yr)


@pytest.mark.gpu
def test_smote_interpolate_is_the_midpoint():
    """'interpolate' (smote.py:26-29) does not run under today's numpy in the reference (np.average of a ragged pair);
    restated as what it computed in 2017, the midpoint of the row and each neighbour.  Parity unpinned."""
    from smrt_amd.balance import smote_balance
    rs = np.random.RandomState(5)
    X = np.vstack([rs.normal(size=(40, 4)), 5 + rs.normal(size=(8, 4))])
    y = np.array([0] * 40 + [1] * 8)
    Xb, yb = smote_balance(X, y, balance_ratio=0.5, strategy="interpolate", n_neighbors=3, random_state=0, shuffle=False)
    Xo, yo = so.smote_balance(X, y, balance_ratio=0.5, strategy="interpolate", n_neighbors=3, random_state=0, shuffle=False)
    assert np.array_equal(Xb, Xo) and np.array_equal(yb, yo)
    minority = X[40:]
    for row in Xb[48:]:
        two = 2.0 * row[None, :] - minority                        # the partner of each candidate row
        assert np.isclose(two[:, None, :], minority[None, :, :], atol=1e-12).all(-1).any()
