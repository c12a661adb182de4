"""Host logic (validation, orchestration, RNG stream order) against fixtures produced by the
reference's own Python (tests/golden/make_host_goldens.py).  Both the oracle restatement and the
product package are held to the same fixtures.  No GPU needed: estimators are duck-typed prefits,
exactly as in the generating script.
"""
import json
import os

import numpy as np
import pytest

from oracle import balance_oracle
from oracle.vae_oracle import topography_shape, resolve_n_latent
from smrt_amd.autoencode import VariationalAutoEncoder
from smrt_amd.autoencode.topography import VAETopography
from smrt_amd.autoencode.vae import _check_activation_optimizer, _bounded_float
from smrt_amd.balance import smrt_balance
from smrt_amd.balance.base import validate_X_y_ratio_classes
from smrt_amd.utils import SeededRandomState, get_random_state

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = np.load(os.path.join(HERE, "golden", "host_goldens.npz"), allow_pickle=False)
with open(os.path.join(HERE, "golden", "host_goldens.json")) as f:
    J = json.load(f)


def exc_name(fn, *a, **k):
    try:
        fn(*a, **k)
    except Exception as e:  # noqa: BLE001
        return type(e).__name__
    return None


class FakeVAE(object):
    def __init__(self, tag, d):
        self.tag, self.d = tag, d

    def generate(self, n=1, **kw):
        base = np.arange(n, dtype=np.float64)[:, None] + np.arange(self.d)[None, :] / 10.0
        return (base + 1000.0 * self.tag + kw.get("loc", 0.0)).astype(np.float32)

    def generate_from_sample(self, X, **kw):
        return (np.asarray(X) * 2.0 + 1000.0 * self.tag + kw.get("loc", 0.0)).astype(np.float32)


VALIDATORS = {"oracle": balance_oracle.validate_X_y_ratio, "product": validate_X_y_ratio_classes}


@pytest.mark.parametrize("impl", sorted(VALIDATORS))
@pytest.mark.parametrize("case", sorted(J["validate"]))
def test_validate_matches_reference(impl, case):
    g = J["validate"][case]
    y = NPZ["val_%s_y" % case]
    X = np.zeros((len(y), 2))
    Xv, yv, n_classes, present, counts, maj, target = VALIDATORS[impl](X, y, g["ratio"])
    assert n_classes == g["n_classes"]
    assert [str(p) for p in present] == g["present"]
    assert [int(c) for c in counts] == g["counts"]
    assert str(maj) == g["majority"]
    assert int(target) == g["target"]
    assert str(Xv.dtype) == g["x_dtype"] and (Xv is not X) == g["x_is_copy"]


@pytest.mark.parametrize("impl", sorted(VALIDATORS))
def test_validate_errors_match_reference(impl):
    v = VALIDATORS[impl]
    rs = np.random.RandomState(0)
    x = rs.rand(202, 4)
    lab101 = np.concatenate([np.arange(101), np.arange(101)])
    single = np.zeros(202)
    single[0] = 1
    got = {
        "too_many_classes": exc_name(v, x, lab101, 0.2),
        "continuous_target": exc_name(v, x, np.linspace(0, 1000, 202), 0.2),
        "single_sample_class": exc_name(v, x, single, 0.2),
        "ratio_zero": exc_name(v, x, (np.arange(202) % 2), 0.0),
        "ratio_gt_one": exc_name(v, x, (np.arange(202) % 2), 1.5),
        "ratio_one_ok": exc_name(v, x, (np.arange(202) % 2), 1.0),
        "y_2d_column_ok": exc_name(v, x, (np.arange(202) % 2)[:, None], 0.5),
    }
    assert got == J["validate_errors"]


def _oracle_balance(X, y, prefit, **kw):
    return balance_oracle.balance(X, y, make_estimator=None, prefit_estimators=prefit, return_estimators=True, **kw)


def _product_balance(X, y, prefit, **kw):
    return smrt_balance(X, y, n_hidden=5, n_latent_factors=2, prefit_estimators=prefit, return_estimators=True, **kw)


BALANCE_CASES = {
    "b_noshuf": ([2], dict(balance_ratio=0.2, shuffle=False)),
    "b_shuf": ([2], dict(balance_ratio=0.2, shuffle=True, random_state=42)),
    "b_samples": ([2], dict(balance_ratio=0.3, shuffle=True, random_state=7, gen_from_samples=True)),
    "b_genargs": ([2], dict(balance_ratio=0.25, shuffle=False, generate_args={"loc": 3.0})),
    "b_str": (["pos"], dict(balance_ratio=0.5, shuffle=True, random_state=3)),
    "b_noop": ([], dict(balance_ratio=0.1, shuffle=False)),
}


@pytest.mark.parametrize("impl", ["oracle", "product"])
@pytest.mark.parametrize("case", sorted(BALANCE_CASES))
def test_balance_orchestration_matches_reference(impl, case):
    labels, kw = BALANCE_CASES[case]
    X, y = NPZ[case + "_X"], NPZ[case + "_y"]
    prefit = {lab: FakeVAE(i + 1, X.shape[1]) for i, lab in enumerate(labels)}
    fn = _oracle_balance if impl == "oracle" else _product_balance
    Xo, yo, enc = fn(X, y, prefit, **kw)
    assert Xo.dtype == np.float64
    np.testing.assert_array_equal(Xo, NPZ[case + "_Xout"])
    np.testing.assert_array_equal(np.asarray(yo).astype(str), NPZ[case + "_yout"].astype(str))
    assert [str(k) for k in enc.keys()] == list(NPZ[case + "_enc_keys"])
    assert [enc[k] is None for k in enc.keys()] == list(NPZ[case + "_enc_none"])


def test_balance_prefit_must_be_dict():
    X, y = NPZ["b_noshuf_X"], NPZ["b_noshuf_y"]
    assert exc_name(smrt_balance, X, y, 5, 2, prefit_estimators=[1, 2]) == J["balance_errors"]["prefit_not_dict"]
    assert exc_name(balance_oracle.balance, X, y, None, prefit_estimators=[1, 2]) == "ValueError"


def test_multi_minority_classes_use_original_labels():
    """The reference raises IndexError here under numpy >= 1.13 (recorded in the fixture); the restated
    behaviour masks the original labels, i.e. what numpy 1.12 computed."""
    assert J["multi_minority_under_this_numpy"] == "IndexError"
    X, y = NPZ["b_noshuf_X"], NPZ["b_noshuf_y"]
    prefit = {1: FakeVAE(1, 3), 2: FakeVAE(2, 3)}
    Xa, ya, _ = _oracle_balance(X, y, prefit, balance_ratio=0.5, shuffle=False)
    Xb, yb, _ = _product_balance(X, y, prefit, balance_ratio=0.5, shuffle=False)
    np.testing.assert_array_equal(Xa, Xb)
    np.testing.assert_array_equal(ya, yb)
    assert (ya == 1).sum() == 20 and (ya == 2).sum() == 20 and (ya == 0).sum() == 40
    # class-1 block then class-2 block, each produced by its own estimator
    assert np.all(Xa[60:65, 0] >= 1000) and np.all(Xa[65:80, 0] >= 2000)


def test_numpy_stream_pins():
    np.testing.assert_array_equal(np.random.RandomState(42).normal(size=(5, 3)), NPZ["rs42_normal_5x3"])
    np.testing.assert_array_equal(np.random.RandomState(42).normal(size=(4, 2), loc=1.5, scale=0.25),
                                  NPZ["rs42_normal_loc_scale"])
    srs = SeededRandomState(77)
    g = J["seeded_random_state"]
    assert srs.seed_value == g["int_seed_value"]
    np.testing.assert_array_equal(srs.state.normal(size=4), NPZ["srs77_first_normals"])
    assert SeededRandomState(None).seed_value == g["none_seed_value"]
    assert (get_random_state(srs) is srs) == g["passthrough_same_object"]
    assert exc_name(SeededRandomState, 1.5) == g["float_seed"]
    assert exc_name(SeededRandomState, "42") == g["str_seed"]
    # documented divergence: the reference's RandomState / numpy-int seeds trip its own int check
    assert g["from_RandomState"] == "TypeError" and g["numpy_int64_seed"] == "TypeError"
    assert SeededRandomState(np.random.RandomState(123)).seed_value == int(np.random.RandomState(123).get_state()[1][0])
    assert SeededRandomState(np.int64(5)).seed_value == 5


def test_constructor_contract():
    vae = VariationalAutoEncoder(n_hidden=450, n_latent_factors=10)
    params = vae.get_params()
    assert set(params) == set(J["vae_default_params"]) and len(params) == 20
    for k, v in J["vae_default_params"].items():
        assert params[k] == v, k
    from sklearn.base import clone
    c = clone(VariationalAutoEncoder(n_hidden=[30, 20], n_latent_factors=0.25, learning_function="adam"))
    assert c.get_params()["n_hidden"] == [30, 20] and c.get_params()["learning_function"] == "adam"


def _validate(**kw):
    v = VariationalAutoEncoder(n_hidden=5, n_latent_factors=2, **kw)
    v._validate_for_fit()
    _bounded_float(v, "eps")


def test_validate_for_fit_matrix():
    values = {
        "min_change": [0.0, 1.0, 1e-3, -1.0, 0.999], "learning_rate": [0.0, 1.0, 0.05, 2.0],
        "dropout": [0.0, 1.0, 0.5, 1.5], "l2_penalty": [None, 0.0, 1.0, 1e-4], "eps": [0.0, 1.0, 1e-10],
        "n_epochs": [0, 1, -1, True, 2.7, "3"], "verbose": [0, 2, -1, False], "display_step": [0, 5, -2],
    }
    for key, vals in values.items():
        got = [[repr(v), exc_name(_validate, **{key: v})] for v in vals]
        assert got == J["validate_for_fit"][key], key


def test_activation_optimizer_errors():
    g = J["activation_optimizer_errors"]
    assert exc_name(_check_activation_optimizer, "swish", "sgd") == g["bad_activation"]
    assert exc_name(_check_activation_optimizer, 3, "sgd") == g["nonstring_activation"]
    assert exc_name(_check_activation_optimizer, "relu", "lbfgs") == g["bad_optimizer"]
    assert exc_name(_check_activation_optimizer, "relu", None) == g["nonstring_optimizer"]
    got = [exc_name(_check_activation_optimizer, a, o)
           for a in ("elu", "identity", "relu", "sigmoid", "tanh")
           for o in ("adadelta", "adagrad", "adagrad-da", "adam", "momentum", "proximal-sgd",
                     "proximal-adagrad", "rms_prop", "sgd")]
    assert got == g["all_ok"]


def _topo(n_hidden, d, z, **kw):
    t = VAETopography(n_hidden=n_hidden, input_shape=d, activation="sigmoid", n_latent_factors=z, **kw)
    return dict(shape=list(t.shape), n_latent=int(t.n_latent_factors),
                enc=[[l.fan_in, l.fan_out] for l in t.encode_layers_],
                dec=[[l.fan_in, l.fan_out] for l in t.decode_layers_])


def test_topography_matches_reference():
    g = J["topography"]
    hidden = [30, 20]
    assert _topo(450, 784, 10) == g["n1"]
    assert _topo(hidden, 64, 8) == g["two_hidden"]
    assert _topo(16, 100, 0.25) == g["float_latent"]
    assert _topo(16, 5, 0.1) == g["float_latent_floor"]
    # documented divergence: the reference leaves D inserted in the caller's list (layer.py:292)
    assert g["caller_list_after_call"] == [64, 30, 20] and hidden == [30, 20]
    for name, t in g.items():
        if isinstance(t, dict):
            d, z = t["shape"][0], t["n_latent"]
            assert list(topography_shape(d, [fo for _, fo in t["enc"]], z)) == t["shape"]
    assert resolve_n_latent(0.25, 100) == 25 and resolve_n_latent(0.1, 5) == 2
    ge = J["topography_errors"]
    assert exc_name(_topo, "450", 10, 2) == ge["n_hidden_str"]
    assert exc_name(_topo, 4.5, 10, 2) == ge["n_hidden_float"]
    assert exc_name(_topo, 5, 10, "2") == ge["latent_str"]
    assert exc_name(_topo, 5, 10, 1.5) == ge["latent_float_ge_1"]
    assert exc_name(_topo, 5, 10, -1) == ge["latent_negative"]
    assert exc_name(_topo, 5, 10, 2, layer_type="he") == ge["bad_layer_type"]
    assert exc_name(_topo, 5, 10, 2, bias_strategy="twos") == ge["bad_bias"]


@pytest.mark.parametrize("dtype", [np.uint64, np.uint8, np.int16])
def test_small_integer_labels_of_any_integer_dtype(dtype):
    """The counting fast path takes unsigned 64-bit labels too (np.bincount refuses them uncast) and agrees with the
    np.unique path the reference takes (smrt/balance/base.py:36-47)."""
    rs = np.random.RandomState(0)
    y = np.concatenate([np.zeros(40), np.ones(7), np.full(5, 3)]).astype(dtype)
    X = rs.uniform(size=(len(y), 4))
    _, yv, n_classes, present, counts, majority, target = validate_X_y_ratio_classes(X, y, 0.5)
    u, c = np.unique(y, return_counts=True)
    assert n_classes == 3 and np.array_equal(present, u) and np.array_equal(counts, c)
    assert majority == 0 and target == 20 and present.dtype == y.dtype


@pytest.mark.parametrize("layer_type", ["xavier", "gaussian"])
def test_initial_parameters_are_the_per_layer_numpy_draws_bit_for_bit(layer_type):
    """``initial_parameters`` takes every layer from ONE shared unit stream (a layer's draws are a prefix of the longest
    layer's, same-shaped layers are computed once): the values must be those of the per-layer definition --
    ``RandomState(seed).uniform(-b, b, (fi, fo))`` / ``.normal(0, sqrt(2 / fi), (fi, fo))`` -- bit for bit, for cached
    and uncached streams, from several threads, and every layer must own its array."""
    from concurrent.futures import ThreadPoolExecutor
    from smrt_amd.autoencode import topography

    def direct(dims, seed):
        out = []
        for fi, fo in dims:
            rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
            if layer_type == "xavier":
                bound = np.sqrt(6.0 / (fi + fo))
                W = rs.uniform(-bound, bound, size=(fi, fo))
            else:
                W = rs.normal(0.0, (2.0 / fi) ** 0.5, size=(fi, fo))
            out.append(W.astype(np.float32))
        return out

    nets = ([(784, 450), (450, 10), (450, 10), (10, 450), (450, 784)],            # N1: the heads share their values
            [(64, 128), (128, 16), (128, 16), (16, 128), (128, 64)],
            [(7, 3), (3, 2), (3, 2), (2, 3), (3, 7)],
            [(5, 5), (5, 5), (5, 5)],
            [(40, 300), (300, 300), (300, 300), (300, 9), (300, 9), (9, 300), (300, 300), (300, 40)])

    def check(args):
        dims, seed = args
        got = topography.initial_parameters(dims, layer_type, "ones", seed)
        want = direct(dims, seed)
        ok = all(w.dtype == np.float32 and np.array_equal(w.view(np.uint32), v.view(np.uint32)) and np.all(b == 1.0)
                 for (w, b), v in zip(got, want))
        return ok and len({id(w) for w, _ in got}) == len(got)

    cases = [(dims, seed) for seed in (42, 0, (1 << 32) + 5) for dims in nets]
    topography._STREAM_CACHE.clear()
    assert all(check(c) for c in cases)                  # growing and shrinking streams, cache hits and misses
    with ThreadPoolExecutor(max_workers=4) as pool:
        assert all(pool.map(check, cases * 2))
