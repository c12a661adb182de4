"""Generate host-logic golden fixtures by running the REFERENCE's own Python.

Run in the build container only (``python tests/golden/make_host_goldens.py``); the
reference never travels to the GPU box, only the small fixtures written here do.

What runs for real: ``smrt.utils``, ``smrt.balance.base``,
``smrt.balance.smrt.smrt_balance`` (with duck-typed *prefit* estimators, so no graph is
built), parameter validation of ``VariationalAutoEncoder`` and the dimension wiring of
``SymmetricalVAETopography``.  TensorFlow is absent, so ``sys.modules['tensorflow']`` is
a MagicMock: nothing numeric that TensorFlow would evaluate is produced or pinned here.
In-memory shims only (np.int/np.bool/np.float aliases, sklearn.externals.six); nothing is
written outside tests/golden/.
"""
import json
import os
import sys
import types
import warnings
from unittest import mock

import numpy as np

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True

import six  # noqa: E402
import sklearn  # noqa: E402
import sklearn.base  # noqa: E402
import sklearn.neighbors  # noqa: E402
import sklearn.preprocessing  # noqa: E402
import sklearn.utils.multiclass  # noqa: E402
import sklearn.utils.validation  # noqa: E402

np.int, np.bool, np.float = int, bool, float
_ext = types.ModuleType("sklearn.externals")
_ext.six = six
sys.modules["sklearn.externals"] = _ext
sys.modules["sklearn.externals.six"] = six
sklearn.externals = _ext
_tf = mock.MagicMock()
_tf.float32 = np.float32
sys.modules["tensorflow"] = _tf
for _m in ("tensorflow.examples", "tensorflow.examples.tutorials", "tensorflow.examples.tutorials.mnist"):
    sys.modules[_m] = mock.MagicMock()
sys.path.insert(0, "/root/reference")

from smrt.autoencode import VariationalAutoEncoder  # noqa: E402
from smrt.autoencode.autoencoder import _validate_activation_optimization  # noqa: E402
from smrt.autoencode.layer import SymmetricalVAETopography  # noqa: E402
from smrt.balance import smrt_balance  # noqa: E402
from smrt.balance.base import _validate_X_y_ratio_classes  # noqa: E402
from smrt.utils import SeededRandomState, get_random_state  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def exc_name(fn, *a, **k):
    try:
        fn(*a, **k)
    except Exception as e:  # noqa: BLE001
        return type(e).__name__
    return None


class FakeVAE(object):
    """Duck-types the two methods smrt_balance calls on a prefit estimator (smrt.py:229-234)."""

    def __init__(self, tag, d):
        self.tag, self.d = tag, d

    def generate(self, n=1, **kw):
        base = np.arange(n, dtype=np.float64)[:, None] + np.arange(self.d)[None, :] / 10.0
        return (base + 1000.0 * self.tag + kw.get("loc", 0.0)).astype(np.float32)

    def generate_from_sample(self, X, **kw):
        return (np.asarray(X) * 2.0 + 1000.0 * self.tag + kw.get("loc", 0.0)).astype(np.float32)


def balance_case(name, X, y, prefit_labels, **kw):
    pre = {lab: FakeVAE(i + 1, X.shape[1]) for i, lab in enumerate(prefit_labels)}
    Xo, yo, enc = smrt_balance(X, y, n_hidden=5, n_latent_factors=2, prefit_estimators=pre,
                               return_estimators=True, **kw)
    return {
        name + "_X": np.asarray(X), name + "_y": np.asarray(y),
        name + "_Xout": Xo, name + "_yout": yo,
        name + "_enc_keys": np.array([str(k) for k in enc.keys()]),
        name + "_enc_none": np.array([enc[k] is None for k in enc.keys()]),
    }


def main():
    out_npz, out_json = {}, {}

    # ---- 1. _validate_X_y_ratio_classes (balance/base.py:19-64)
    rs = np.random.RandomState(0)
    cases = {
        "mnist_counts": (np.zeros((5944, 2)), np.array([1] * 500 + [0] * 5444), 0.5),
        "three_class": (rs.rand(60, 3), np.array([0] * 40 + [1] * 15 + [2] * 5), 0.2),
        "string_labels": (rs.rand(30, 2), np.array(["a"] * 20 + ["b"] * 6 + ["c"] * 4), 1.0),
        "tie_first_max": (rs.rand(20, 2), np.array([3] * 8 + [1] * 8 + [2] * 4), 0.9),
        "tiny_ratio": (rs.rand(12, 2), np.array([0] * 9 + [1] * 3), 0.01),
    }
    val = {}
    for k, (X, y, ratio) in cases.items():
        Xv, yv, n_classes, present, counts, maj, target = _validate_X_y_ratio_classes(X, y, ratio)
        val[k] = dict(ratio=ratio, n_classes=int(n_classes), present=[str(p) for p in present],
                      counts=[int(c) for c in counts], majority=str(maj), target=int(target),
                      x_dtype=str(Xv.dtype), x_is_copy=bool(Xv is not X))
        out_npz["val_%s_y" % k] = np.asarray(y)
    out_json["validate"] = val
    x = rs.rand(202, 4)
    lab101 = np.concatenate([np.arange(101), np.arange(101)])
    single = np.zeros(202)
    single[0] = 1
    out_json["validate_errors"] = {
        "too_many_classes": exc_name(_validate_X_y_ratio_classes, x, lab101, 0.2),
        "continuous_target": exc_name(_validate_X_y_ratio_classes, x, np.linspace(0, 1000, 202), 0.2),
        "single_sample_class": exc_name(_validate_X_y_ratio_classes, x, single, 0.2),
        "ratio_zero": exc_name(_validate_X_y_ratio_classes, x, (np.arange(202) % 2), 0.0),
        "ratio_gt_one": exc_name(_validate_X_y_ratio_classes, x, (np.arange(202) % 2), 1.5),
        "ratio_one_ok": exc_name(_validate_X_y_ratio_classes, x, (np.arange(202) % 2), 1.0),
        "y_2d_column_ok": exc_name(_validate_X_y_ratio_classes, x, (np.arange(202) % 2)[:, None], 0.5),
    }

    # ---- 2. smrt_balance orchestration with prefit fakes (balance/smrt.py:170-250)
    X3 = rs.rand(60, 3)
    y3 = np.array([0] * 40 + [1] * 15 + [2] * 5)
    perm = rs.permutation(60)
    X3, y3 = X3[perm], y3[perm]
    # NOTE: smrt.py:207 masks the ORIGINAL X with the GROWING y_transform (smrt.py:238), so a second
    # class that needs synthesis raises IndexError under numpy>=1.13 (it was only a DeprecationWarning
    # under the reference's numpy 1.12, where the extra mask entries -- all False -- were ignored).
    # Goldens therefore use inputs where exactly ONE class needs synthesis; the multi-class
    # behaviour (mask the original labels) is restated and documented in DESIGN.md.
    out_npz.update(balance_case("b_noshuf", X3, y3, [2], balance_ratio=0.2, shuffle=False))
    out_npz.update(balance_case("b_shuf", X3, y3, [2], balance_ratio=0.2, shuffle=True, random_state=42))
    out_npz.update(balance_case("b_samples", X3, y3, [2], balance_ratio=0.3, shuffle=True,
                                random_state=7, gen_from_samples=True))
    out_npz.update(balance_case("b_genargs", X3, y3, [2], balance_ratio=0.25, shuffle=False,
                                generate_args={"loc": 3.0}))
    ys = np.array(["neg"] * 30 + ["pos"] * 6)
    Xs = rs.rand(36, 2)
    out_npz.update(balance_case("b_str", Xs, ys, ["pos"], balance_ratio=0.5, shuffle=True, random_state=3))
    # nothing to do: every class already at/above target -> encoders all None
    out_npz.update(balance_case("b_noop", X3, y3, [], balance_ratio=0.1, shuffle=False))
    out_json["multi_minority_under_this_numpy"] = exc_name(
        balance_case, "x", X3, y3, [1, 2], balance_ratio=0.5, shuffle=False)
    out_json["balance_errors"] = {
        "prefit_not_dict": exc_name(smrt_balance, X3, y3, 5, 2, prefit_estimators=[1, 2]),
    }

    # ---- 3. numpy stream pins (autoencoder.py:571, utils.py:41-53)
    out_npz["rs42_normal_5x3"] = np.random.RandomState(42).normal(size=(5, 3))
    out_npz["rs42_normal_loc_scale"] = np.random.RandomState(42).normal(size=(4, 2), loc=1.5, scale=0.25)
    srs = SeededRandomState(77)
    out_json["seeded_random_state"] = {
        # utils.py:46-50: the first MT word is a numpy.uint32, which then fails the int check
        "from_RandomState": exc_name(SeededRandomState, np.random.RandomState(123)),
        "int_seed_value": int(srs.seed_value),
        "none_seed_value": int(SeededRandomState(None).seed_value),
        "passthrough_same_object": bool(get_random_state(srs) is srs),
        "float_seed": exc_name(SeededRandomState, 1.5),
        "str_seed": exc_name(SeededRandomState, "42"),
        "numpy_int64_seed": exc_name(SeededRandomState, np.int64(5)),
    }
    out_npz["srs77_first_normals"] = srs.state.normal(size=4)

    # ---- 4. constructor / validation contract (base.py:79-90, autoencoder.py:51-88,416-446)
    vae = VariationalAutoEncoder(n_hidden=450, n_latent_factors=10)
    params = vae.get_params()
    out_json["vae_default_params"] = {k: (v if not isinstance(v, float) else float(v)) for k, v in params.items()}

    def validate(**kw):
        v = VariationalAutoEncoder(n_hidden=5, n_latent_factors=2, **kw)
        v._validate_for_fit()
        from smrt.autoencode.base import _validate_float
        _validate_float(v, "eps")

    matrix = {}
    for key, values in {
        "min_change": [0.0, 1.0, 1e-3, -1.0, 0.999],
        "learning_rate": [0.0, 1.0, 0.05, 2.0],
        "dropout": [0.0, 1.0, 0.5, 1.5],
        "l2_penalty": [None, 0.0, 1.0, 1e-4],
        "eps": [0.0, 1.0, 1e-10],
        "n_epochs": [0, 1, -1, True, 2.7, "3"],
        "verbose": [0, 2, -1, False],
        "display_step": [0, 5, -2],
    }.items():
        matrix[key] = [[repr(v), exc_name(validate, **{key: v})] for v in values]
    out_json["validate_for_fit"] = matrix
    out_json["activation_optimizer_errors"] = {
        "bad_activation": exc_name(_validate_activation_optimization, "swish", "sgd"),
        "nonstring_activation": exc_name(_validate_activation_optimization, 3, "sgd"),
        "bad_optimizer": exc_name(_validate_activation_optimization, "relu", "lbfgs"),
        "nonstring_optimizer": exc_name(_validate_activation_optimization, "relu", None),
        "all_ok": [exc_name(_validate_activation_optimization, a, o)
                   for a in ("elu", "identity", "relu", "sigmoid", "tanh")
                   for o in ("adadelta", "adagrad", "adagrad-da", "adam", "momentum", "proximal-sgd",
                             "proximal-adagrad", "rms_prop", "sgd")],
    }

    # ---- 5. topography wiring (layer.py:246-345) -- dimensions only, tensors are mocks
    def topo(n_hidden, d, z, **kw):
        t = SymmetricalVAETopography(X_placeholder=mock.MagicMock(), n_hidden=n_hidden, input_shape=d,
                                     activation=mock.MagicMock(), n_latent_factors=z, **kw)
        return dict(shape=list(t.shape), n_latent=int(t.n_latent_factors),
                    enc=[[l.fan_in, l.fan_out] for l in t.encode_layers_],
                    dec=[[l.fan_in, l.fan_out] for l in t.decode_layers_])

    hidden_list = [30, 20]
    out_json["topography"] = {
        "n1": topo(450, 784, 10),
        "two_hidden": topo(hidden_list, 64, 8),
        "float_latent": topo(16, 100, 0.25),
        "float_latent_floor": topo(16, 5, 0.1),
        "caller_list_after_call": hidden_list,       # layer.py:292 mutates the caller's list
    }
    out_json["topography_errors"] = {
        "n_hidden_str": exc_name(topo, "450", 10, 2),
        "n_hidden_float": exc_name(topo, 4.5, 10, 2),
        "latent_str": exc_name(topo, 5, 10, "2"),
        "latent_float_ge_1": exc_name(topo, 5, 10, 1.5),
        "latent_negative": exc_name(topo, 5, 10, -1),
        "bad_layer_type": exc_name(topo, 5, 10, 2, layer_type="he"),
        "bad_bias": exc_name(topo, 5, 10, 2, bias_strategy="twos"),
    }

    np.savez_compressed(os.path.join(HERE, "host_goldens.npz"), **out_npz)
    with open(os.path.join(HERE, "host_goldens.json"), "w") as f:
        json.dump(out_json, f, indent=1, sort_keys=True, default=str)
    print("wrote", len(out_npz), "arrays;", list(out_json))


if __name__ == "__main__":
    main()
