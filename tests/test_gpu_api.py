"""The drop-in Python surface on the GPU: VariationalAutoEncoder and smrt_balance.

Structural contracts restate the reference's own tests (smrt/autoencode/tests/test_vae.py:19-54,
smrt/balance/tests/test_smrt.py:12-37) on MNIST-*shaped* synthetic data (real MNIST is a network
download there); numeric checks compare end-to-end results with the CPU oracle.
"""
import numpy as np
import pytest
from sklearn.exceptions import NotFittedError

from oracle import balance_oracle
from oracle.vae_oracle import OracleVAE, init_params

pytestmark = pytest.mark.gpu


def mnist_like(n_major, n_minor, d=784, seed=0):
    """Per-class prototype in [0,1]^d, ~80 % exact zeros (SURVEY 8d stand-in for load_imbalanced_mnist)."""
    rs = np.random.RandomState(seed)
    Xs, ys = [], []
    for label, n in ((1, n_minor), (0, n_major)):          # minority rows first, as the reference loader stacks them
        proto = rs.uniform(0, 1, size=d) * (rs.uniform(size=d) > 0.8)
        X = np.clip(proto + 0.15 * rs.normal(size=(n, d)), 0, 1) * (proto > 0)
        Xs.append(X)
        ys.append(np.full(n, label))
    return np.vstack(Xs), np.concatenate(ys)


class OracleEstimator(object):
    """smrt_balance-facing adapter around the oracle VAE (same duck type as the product estimator)."""

    def __init__(self, X_sub, seeded_rs, seed, n_hidden, n_latent, **kw):
        D = X_sub.shape[1]
        self.n_latent_factors = n_latent
        self.rs = seeded_rs
        fit_kw = {k: kw.pop(k) for k in ("n_epochs", "batch_size")}
        self.o = OracleVAE(D, n_hidden, n_latent, seed=seed, **kw)
        self.o.fit(X_sub, **fit_kw)

    def generate(self, n=1, **nrm):
        return self.o.generate(self.rs, n, **nrm)

    def generate_from_sample(self, X, **nrm):
        return self.o.generate_from_sample(self.rs, X, **nrm)


def test_vae_contract_like_reference_test():
    from smrt_amd.autoencode import VariationalAutoEncoder
    X, _ = mnist_like(700, 0, seed=1)
    X_train, X_test = X[:500], X[500:]
    ae = VariationalAutoEncoder(n_hidden=400, n_latent_factors=20, n_epochs=10, learning_rate=0.01, batch_size=256,
                                display_step=5, activation_function='sigmoid', verbose=2, random_state=42,
                                layer_type='gaussian')
    with pytest.raises(NotFittedError):
        ae.generate()
    ae.fit(X_train)
    assert ae.topography_.shape == (784, 20, 400, 784)
    assert len(ae.epoch_costs_) == len(ae.epoch_times_) == 10 and np.isfinite(ae.train_cost_)
    assert ae.train_cost_ == ae.epoch_costs_[-1]
    assert ae.epoch_costs_[-1] < ae.epoch_costs_[0]                      # it learns
    enc = ae.transform(X_train)
    assert isinstance(enc, list) and len(enc) == 2
    assert enc[0].shape == enc[1].shape == (500, 20) and enc[0].dtype == np.float32
    g = ae.generate()
    assert g.shape == (1, 784) and g.dtype == np.float32 and np.all((g >= 0) & (g <= 1))
    synth = ae.generate_from_sample(X_test[:5])
    assert synth.shape[0] == 5
    w, b = ae.topography_.get_weights_biases()
    assert [x.shape for x in w] == [(784, 400), (20, 400), (400, 784)] and len(b) == 3
    ae.fit(X_train[:300])                                                 # refit replaces the session
    assert ae.transform(X_test)[0].shape == (200, 20)
    ae.clean_session()
    with pytest.raises(AttributeError):
        ae.transform(X_test)


def test_vae_error_behaviour_on_fit():
    from smrt_amd.autoencode import VariationalAutoEncoder
    X = np.random.RandomState(0).uniform(size=(40, 12))
    with pytest.raises(UnboundLocalError):                                # base.py:243
        VariationalAutoEncoder(5, 2, n_epochs=0).fit(X)
    with pytest.raises(ZeroDivisionError):                                # base.py:225
        VariationalAutoEncoder(5, 2, n_epochs=2, display_step=0).fit(X)
    for opt in ("momentum", "adagrad-da"):                                # base.py:138,141
        with pytest.raises(TypeError):
            VariationalAutoEncoder(5, 2, learning_function=opt).fit(X)
    with pytest.raises(ValueError):
        VariationalAutoEncoder(5, 2, learning_rate=1.0).fit(X)
    with pytest.raises(ValueError):
        VariationalAutoEncoder(5, 2).fit(np.array([[np.nan] * 4] * 4))
    hidden = [7, 5]
    v = VariationalAutoEncoder(hidden, 0.25, n_epochs=1).fit(X)
    assert hidden == [7, 5] and v.topography_.n_latent_factors == 3 and v.topography_.shape == (12, 7, 3, 5, 7, 12)
    with pytest.raises(TypeError):                                        # float latent breaks generate (autoencoder.py:571)
        v.generate(2)
    with pytest.raises(ValueError):
        v.feed_forward(np.zeros((2, 9)))


@pytest.mark.parametrize("opt,act", [("rms_prop", "sigmoid"), ("adam", "tanh"), ("sgd", "relu")])
def test_fit_generate_match_oracle(opt, act):
    from smrt_amd.autoencode import VariationalAutoEncoder
    rs = np.random.RandomState(3)
    X = rs.uniform(0, 1, size=(300, 30))
    kw = dict(learning_rate=0.01, l2_penalty=1e-4)
    ae = VariationalAutoEncoder(n_hidden=[24, 12], n_latent_factors=5, n_epochs=3, batch_size=128, random_state=9,
                                learning_function=opt, activation_function=act, **kw).fit(X)
    o = OracleVAE(30, [24, 12], 5, activation=act, learning_function=opt, seed=9, **kw)
    o.fit(X, n_epochs=3, batch_size=128)
    np.testing.assert_allclose(ae.epoch_costs_, np.array(o.epoch_costs_), rtol=1e-4)
    for (W, b), (Wo, bo) in zip(ae.get_weights(), o.params):
        np.testing.assert_allclose(W, Wo, rtol=1e-4, atol=1e-4 * np.abs(Wo).max())
        np.testing.assert_allclose(b, bo, rtol=1e-4, atol=1e-4 * max(np.abs(bo).max(), np.abs(Wo).max()))
    o_rs = np.random.RandomState(9)                                       # generate() consumes random_state_
    np.testing.assert_allclose(ae.generate(7), o.generate(o_rs, 7), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ae.generate(3, loc=0.5, scale=2.0), o.generate(o_rs, 3, loc=0.5, scale=2.0),
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ae.generate_from_sample(X[:11]), o.generate_from_sample(o_rs, X[:11]),
                               rtol=1e-4, atol=1e-5)
    mu, ls = ae.transform(X[:50])
    mu_o, ls_o = o.encode(X[:50])
    np.testing.assert_allclose(mu, mu_o, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ls, ls_o, rtol=1e-4, atol=1e-5)


def test_weight_injection_and_checkpoint_roundtrip():
    from smrt_amd.autoencode import VariationalAutoEncoder
    rs = np.random.RandomState(4)
    X = rs.uniform(0, 1, size=(64, 10))
    params = init_params(10, [6], 3, layer_type="gaussian", seed=123)
    params = [(W + 0.1 * rs.normal(size=W.shape).astype(np.float32), b + 0.1) for W, b in params]
    ae = VariationalAutoEncoder(6, 3, n_epochs=2, batch_size=32, random_state=1, learning_function="sgd",
                                learning_rate=0.01).set_initial_weights(params).fit(X)
    o = OracleVAE(10, [6], 3, learning_function="sgd", learning_rate=0.01, seed=1, params=params)
    o.fit(X, n_epochs=2, batch_size=32)
    np.testing.assert_allclose(ae.epoch_costs_, np.array(o.epoch_costs_), rtol=1e-4)
    saved = ae.get_weights()
    z = rs.normal(size=(4, 3))
    before = ae.feed_forward(z)
    ae.set_weights([(W * 0, b * 0) for W, b in saved])
    assert np.allclose(ae.feed_forward(z), 0.5)
    ae.set_weights(saved)
    np.testing.assert_array_equal(ae.feed_forward(z), before)


def test_smrt_balance_counts_like_reference_test():
    """test_smrt.py:12-37 on the MNIST-shaped stand-in: 5444 zeros + 500 ones -> 8166 rows, 2722 ones."""
    from smrt_amd.balance import smrt_balance
    X, y = mnist_like(5444, 500, seed=2)
    assert X.shape[0] == y.shape[0] == 5944

    def _fit(prefit_estimators=None, return_estimators=False, gen_from_samples=False):
        return smrt_balance(X, y, n_hidden=450, n_latent_factors=10, learning_rate=0.05,
                            activation_function='sigmoid', verbose=0, display_step=25, n_epochs=50, batch_size=256,
                            random_state=42, shuffle=False, balance_ratio=0.5, return_estimators=return_estimators,
                            prefit_estimators=prefit_estimators, gen_from_samples=gen_from_samples)

    def _assert(x, yy):
        assert x.shape[0] == yy.shape[0] == 8166 and x.dtype == np.float64
        assert yy[yy == 1].shape[0] == x[yy == 1].shape[0] == 2722
        assert yy[yy == 0].shape[0] == x[yy == 0].shape[0] == 5444
        assert np.all(np.isfinite(x)) and x.min() >= 0 and x.max() <= 1
        np.testing.assert_array_equal(x[:5944], X)                       # originals first when shuffle=False

    X_smrt, y_smrt, encoders = _fit(return_estimators=True)
    _assert(X_smrt, y_smrt)
    assert encoders[0] is None and encoders[1] is not None
    _assert(*_fit(prefit_estimators=encoders))
    _assert(*_fit(gen_from_samples=True))


@pytest.mark.parametrize("gen_from_samples", [False, True])
def test_smrt_balance_matches_oracle_end_to_end(gen_from_samples):
    """Three minority classes, shuffled output: every synthetic row equals the oracle's (same numpy stream
    order, same Philox epsilon stream, same initial weights)."""
    from smrt_amd.balance import smrt_balance
    rs = np.random.RandomState(5)
    D = 20
    counts = {0: 120, 1: 30, 2: 12, 3: 50}
    Xs, ys = [], []
    for lab, n in counts.items():
        c = rs.uniform(0.2, 0.8, size=D)
        Xs.append(np.clip(c + 0.1 * rs.normal(size=(n, D)), 0, 1))
        ys.append(np.full(n, lab))
    perm = rs.permutation(sum(counts.values()))
    X, y = np.vstack(Xs)[perm], np.concatenate(ys)[perm]
    kw = dict(learning_rate=0.01, l2_penalty=1e-4)
    Xb, yb, enc = smrt_balance(X, y, n_hidden=16, n_latent_factors=4, balance_ratio=0.5, n_epochs=3, batch_size=16,
                               random_state=11, shuffle=True, gen_from_samples=gen_from_samples,
                               return_estimators=True, generate_args={"scale": 0.5}, **kw)

    def make(X_sub, rstate):
        return OracleEstimator(X_sub, rstate, 11, [16], 4, n_epochs=3, batch_size=16, **kw)

    Xo, yo = balance_oracle.balance(X, y, make, balance_ratio=0.5, random_state=11, shuffle=True,
                                    gen_from_samples=gen_from_samples, generate_args={"scale": 0.5})
    np.testing.assert_array_equal(yb, yo)
    assert Xb.shape == Xo.shape == (120 + 60 + 60 + 60, D)
    np.testing.assert_allclose(Xb, Xo, rtol=1e-4, atol=1e-5)
    assert enc[0] is None and all(enc[k] is not None for k in (1, 2, 3))


@pytest.mark.gpu
def test_smrt_balance_concurrent_class_fits_are_bit_identical(monkeypatch):
    """Fits of several minority classes run concurrently (one host thread + HIP stream each); the result must equal the
    sequential run bit for bit, estimators included (SURVEY 8e-1 applied inside one GPU)."""
    from smrt_amd.balance import smrt_balance
    rs = np.random.RandomState(8)
    counts = {0: 400, 1: 60, 2: 45, 3: 90, 4: 30, 5: 250}
    X = np.vstack([np.clip(rs.uniform(size=24) + 0.1 * rs.normal(size=(n, 24)), 0, 1) for n in counts.values()])
    y = np.concatenate([np.full(n, k) for k, n in counts.items()])
    p = rs.permutation(len(y))
    X, y = X[p], y[p]
    kw = dict(n_hidden=[20, 12], n_latent_factors=4, balance_ratio=0.5, n_epochs=4, batch_size=32, random_state=3,
              return_estimators=True)
    monkeypatch.setenv("SMRT_CLASS_THREADS", "1")
    Xs, ys, es = smrt_balance(X, y, **kw)
    monkeypatch.setenv("SMRT_CLASS_THREADS", "8")
    Xt, yt, et = smrt_balance(X, y, **kw)
    assert np.array_equal(Xs, Xt) and np.array_equal(ys, yt)
    assert (ys == 1).sum() == 200 and (ys == 4).sum() == 200 and (ys == 5).sum() == 250
    for label in (1, 2, 3, 4):
        np.testing.assert_array_equal(es[label].sess.get_params(), et[label].sess.get_params())
        np.testing.assert_array_equal(es[label].epoch_costs_, et[label].epoch_costs_)
    assert es[0] is None and es[5] is None


@pytest.mark.gpu
def test_diagnostic_peaks_are_sane():
    """include/smrt_hip.h diagnostics: measured fp32 MFMA rate, HBM bandwidth and the empty-step floor land where an
    MI355X can (nominal 157.3 TFLOP/s, 8 TB/s); the step-floor chain checks that every launch ran."""
    from smrt_amd import _hip
    tf = _hip.diag_mfma_peak(0)
    assert 60.0 < tf < 1.02 * 157.3, tf
    copy, read = _hip.diag_hbm_bandwidth(0, 1 << 28)
    assert 1000.0 < copy < 8000.0 and 1000.0 < read < 8000.0, (copy, read)
    us = _hip.diag_step_floor(0, 2, 50, 8, 512)
    assert 0.5 < us < 100.0, us
    with pytest.raises(_hip.HipLibraryError):
        _hip.diag_step_floor(0, 0, 50, 8, 512)


@pytest.mark.gpu
def test_grouped_class_fits_are_bit_identical_to_separate_fits(monkeypatch):
    """smrt_vae_train_epochs_multi: the per-class VAEs of one smrt_balance call trained side by side with grouped launches
    (one launch per kernel for ALL classes) against the same call with every class fitted on its own.  Same kernels on the
    same numbers: outputs, weights and epoch costs must agree bit for bit -- with classes of different sizes (different
    numbers of mini-batches, ragged tails of 44, 2 and 13 rows), a net off the LDS-resident path, Adam."""
    from smrt_amd.balance import smrt_balance
    rs = np.random.RandomState(12)
    D = 200
    counts = {0: 900, 1: 300, 2: 130, 3: 77, 4: 64}
    X = np.vstack([np.clip(rs.uniform(size=D) + 0.1 * rs.normal(size=(n, D)), 0, 1) for n in counts.values()])
    y = np.concatenate([np.full(n, k) for k, n in counts.items()])
    p = rs.permutation(len(y))
    X, y = X[p], y[p]
    kw = dict(n_hidden=300, n_latent_factors=7, balance_ratio=0.5, n_epochs=3, batch_size=64, random_state=3,
              learning_function="adam", learning_rate=0.002, return_estimators=True)
    monkeypatch.setenv("SMRT_CLASS_GROUP", "0")
    Xs, ys, es = smrt_balance(X, y, **kw)
    monkeypatch.setenv("SMRT_CLASS_GROUP", "1")
    Xg, yg, eg = smrt_balance(X, y, **kw)
    assert np.array_equal(Xs, Xg) and np.array_equal(ys, yg)
    for label in (1, 2, 3, 4):
        np.testing.assert_array_equal(es[label].sess.get_params(), eg[label].sess.get_params())
        np.testing.assert_array_equal(es[label].epoch_costs_, eg[label].epoch_costs_)
        assert eg[label].sess.step_count == es[label].sess.step_count == 3 * ((counts[label] + 63) // 64)
    # grouped: one launch per kernel and step position for all four classes, not four
    assert eg[1].sess.last_epoch_launches < 2 * es[1].sess.last_epoch_launches
    o = OracleVAE(D, [300], 7, learning_function="adam", learning_rate=0.002, seed=3)
    o.fit(X[y == 2], n_epochs=3, batch_size=64)
    np.testing.assert_allclose(eg[2].epoch_costs_, np.array(o.epoch_costs_), rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["layers", "fused", "persistent"])
def test_epoch_ring_reports_the_same_costs_as_events_and_copies(monkeypatch, path):
    """The launch that closes an epoch writes the epoch's cost and a time stamp (after a persistent launch also its error
    word) straight into pinned host memory (the epoch ring, FinalizeArgs::ring_*; SMRT_HIP_EPOCH_RING=0: an event pair and
    4-byte copies per epoch).  Both forms must report the same costs bit for bit and leave the same weights on every
    training path (per-layer launches, the two-kernel step, the persistent kernel) -- one call, a second call with
    more epochs (the ring is re-allocated, the cached epoch graph dropped), early stopping, the grouped fit of several
    handles -- and the ring's epoch times must be positive and add up to no more than the call's wall time."""
    import time
    from smrt_amd import _hip
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1" if path == "persistent" else "0")
    monkeypatch.setenv("SMRT_HIP_FUSED", "0" if path == "layers" else "1")
    rs = np.random.RandomState(31)
    D, hidden, Z, B = 64, [128], 16, 128
    X = rs.uniform(0, 1, size=(500, D))

    def mk(opt, lr, seed):
        return _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer=opt, learning_rate=lr,
                              l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=seed,
                              max_batch=B, device=0)

    out = {}
    for ring in ("1", "0"):
        monkeypatch.setenv("SMRT_HIP_EPOCH_RING", ring)
        h = mk("rms_prop", 0.01, 9)
        c1, _ = h.train_epochs(X, B, 3)
        t0 = time.perf_counter()
        c2, ms2 = h.train_epochs(X, B, 40)
        wall_ms = (time.perf_counter() - t0) * 1e3
        c3, _ = h.train_epochs(X[:300], B, 30, early_stopping=True, min_change=1.0)
        assert len(c1) == 3 and len(c2) == 40 and 2 <= len(c3) < 30
        assert (h.last_epoch_launches == 2) == (path == "persistent")
        assert np.all(np.asarray(ms2) > 0.0) and float(np.sum(ms2)) <= wall_ms, (ms2, wall_ms)
        out[ring] = (np.concatenate([c1, c2, c3]), h.get_params(), h.step_count)
        h.close()
    np.testing.assert_array_equal(out["1"][0], out["0"][0])
    np.testing.assert_array_equal(out["1"][1], out["0"][1])
    assert out["1"][2] == out["0"][2]
    if path != "layers":
        return                                           # (grouped fits are for nets off the LDS-resident path)
    grouped = {}
    for ring in ("1", "0"):
        monkeypatch.setenv("SMRT_HIP_EPOCH_RING", ring)
        hs = [mk("adam", 0.002, 9 + k) for k in range(3)]
        Xs = [_hip.to_device_padded(x, hs[0].device) for x in (X[:500], X[:130], X[100:364])]
        res = _hip.train_epochs_multi(hs, Xs, B, 4)
        assert res is not None
        costs, ms = res
        assert np.all(np.asarray(ms) > 0.0)
        grouped[ring] = (np.asarray(costs), [h.get_params() for h in hs])
        for h in hs:
            h.close()
    np.testing.assert_array_equal(grouped["1"][0], grouped["0"][0])
    for a, b in zip(grouped["1"][1], grouped["0"][1]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("D,hidden,Z,B,act,opt", [(784, [450], 10, 128, "sigmoid", "rms_prop"),     # the reference's own net and batch
                                                  (203, [77, 45], 6, 83, "tanh", "adagrad"),       # ragged everything, two hidden layers
                                                  (90, [300], 16, 256, "relu", "sgd")])
def test_small_batch_tile_shapes_agree_bit_for_bit(monkeypatch, D, hidden, Z, B, act, opt):
    """small_gemm.hip picks the tile of a small-batch launch -- 32x32, 16x32 or 16x16 outputs per workgroup -- by how many
    workgroups the grid then has; every shape sums a wave's k-blocks in the same order, so WHICH shape ran must not show
    in any number: three steps (injected epsilon, ragged last batch) with each shape forced, then cost, every gradient,
    the weights and encode / decode compared bit for bit; and against the oracle at 1e-4."""
    from smrt_amd import _hip
    from oracle.vae_oracle import flatten_params
    monkeypatch.setenv("SMRT_HIP_FUSED", "0")
    rs = np.random.RandomState(D)
    o = OracleVAE(D, hidden, Z, activation=act, learning_function=opt, learning_rate=0.01, seed=4)
    o.params = [((W + 0.05 * rs.normal(size=W.shape)).astype(np.float32), (b + 0.05 * rs.normal(size=b.shape)).astype(np.float32))
                for W, b in o.params]
    p0 = flatten_params(o.params).copy()
    steps = [(rs.uniform(0, 1, size=(n, D)), rs.normal(size=(n, Z))) for n in (B, B, max(B // 3, 1))]
    got = {}
    for shape in ("22", "12", "11"):
        monkeypatch.setenv("SMRT_HIP_SMALL_TILE", shape)
        h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation=act, optimizer=opt, learning_rate=0.01,
                           l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=4,
                           max_batch=B, device=0)
        h.set_params(p0)
        outs = []
        for X, eps in steps:
            c, g = h.train_step(X, eps, want_grads=True)
            outs += [np.float32(c), g]
        mu, ls = h.encode(steps[0][0][:50])
        outs += [h.get_params(), mu, ls, h.decode(steps[0][1][:50])]
        got[shape] = outs
        h.close()
    for shape in ("12", "11"):
        for k, (a, b) in enumerate(zip(got["22"], got[shape])):
            np.testing.assert_array_equal(a, b, err_msg="tile shape %s differs from 32x32 in output %d" % (shape, k))
    for k, (X, eps) in enumerate(steps):
        c_o = o.train_step(X, eps)
        assert np.isclose(got["11"][2 * k], c_o, rtol=1e-4), (k, got["11"][2 * k], c_o)
    np.testing.assert_allclose(got["11"][6], flatten_params(o.params), rtol=1e-4, atol=1e-5)
