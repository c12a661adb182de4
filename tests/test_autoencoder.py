"""Plain AutoEncoder (reference smrt/autoencode/autoencoder.py:91-282): the oracle's backward against
torch autograd on CPU, and on the GPU the HIP path (SMRT_MODEL_AE through the C ABI) against the oracle, plus the
estimator contract the reference's own test exercises (smrt/autoencode/tests/test_autoencoder.py:19-50: fit,
topography shape, transform, feed_forward shapes)."""
import numpy as np
import pytest
import torch

from oracle.ae_oracle import OracleAE, ae_layer_dims
from oracle.vae_oracle import ACTIVATIONS, flatten_params

TORCH_ACT = {"sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu,
             "elu": torch.nn.functional.elu, "identity": lambda v: v}


@pytest.mark.parametrize("act", ACTIVATIONS)
@pytest.mark.parametrize("hidden,l2", [([7], 1e-4), ([9, 6], None), ([8, 5, 3], 1e-2)])
def test_ae_backward_matches_autograd(act, hidden, l2):
    rs = np.random.RandomState(0)
    D, B = 11, 6
    X = rs.uniform(0, 1, size=(B, D))
    o = OracleAE(D, hidden, activation=act, l2_penalty=l2, dtype=np.float64, layer_type="gaussian", seed=3)
    o.params = [(W + 0.1 * rs.normal(size=W.shape), b + 0.1 * rs.normal(size=b.shape)) for W, b in o.params]
    outs = o.forward(X)
    c = o.cost(outs)
    gW, gb = o.gradients(outs)
    tp = [(torch.tensor(W, requires_grad=True), torch.tensor(b, requires_grad=True)) for W, b in o.params]
    h = torch.tensor(X)
    for W, b in tp:
        h = TORCH_ACT[act](h @ W + b)
    tc = ((torch.tensor(X) - h) ** 2).mean()
    if l2 is not None:
        tc = tc + l2 * sum(0.5 * (W ** 2).sum() for W, _ in tp)
    tc.backward()
    assert np.isclose(float(tc.detach()), float(c), rtol=1e-12)
    for (W, b), a, d in zip(tp, gW, gb):
        np.testing.assert_allclose(a, W.grad.numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(d, b.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_ae_topography_and_params_contract():
    from smrt_amd.autoencode import AETopography, AutoEncoder
    t = AETopography([400, 60], 784, "sigmoid")
    enc, dec = ae_layer_dims(784, [400, 60])
    assert t.dims == enc + dec and t.shape == (784, 400, 60, 400, 784)
    assert [l.fan_out for l in t.encode_layers_] == [400, 60] and [l.fan_out for l in t.decode_layers_] == [400, 784]
    p = AutoEncoder(n_hidden=400).get_params()
    assert len(p) == 18 and p["learning_function"] == "rms_prop" and p["l2_penalty"] == 1e-4 and p["dropout"] == 1.0
    from sklearn.base import clone
    assert clone(AutoEncoder(n_hidden=[5, 3], n_epochs=7)).n_epochs == 7


# ---------------------------------------------------------------------------------------- GPU
def _close(a, b, what, rtol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30), err_msg=what)


def _handle(D, hidden, act, opt, lr, l2, clip, max_batch):
    from smrt_amd import _hip
    return _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=0, activation=act, optimizer=opt, learning_rate=lr,
                          l2_penalty=l2, clip=clip, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=1,
                          max_batch=max_batch, device=0, model="ae")


@pytest.mark.gpu
@pytest.mark.parametrize("act", ACTIVATIONS)
@pytest.mark.parametrize("opt,l2,clip", [("rms_prop", 1e-4, True), ("adam", None, False), ("sgd", 1e-2, True)])
def test_ae_three_steps_match_oracle(act, opt, l2, clip):
    rs = np.random.RandomState(2)
    D, hidden, B = 23, [17, 6], 37
    o = OracleAE(D, hidden, activation=act, learning_function=opt, learning_rate=0.02, l2_penalty=l2, clip=clip, seed=1)
    o.params = [((W + 0.05 * rs.normal(size=W.shape)).astype(np.float32), (b + 0.05 * rs.normal(size=b.shape)).astype(np.float32))
                for W, b in o.params]
    h = _handle(D, hidden, act, opt, 0.02, l2, clip, B)
    assert h.n_params == flatten_params(o.params).size
    h.set_params(flatten_params(o.params))
    for step in range(3):
        X = rs.uniform(0, 1, size=(B if step < 2 else 9, D))
        c_o, (gW, gb) = o.train_step(X, return_grads=True)
        c_h, g_h = h.train_step(X, want_grads=True)
        assert np.isclose(c_h, c_o, rtol=1e-4), (step, c_h, c_o)
        _close(g_h, np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(gW, gb)]), "gradients")
    _close(h.get_params(), flatten_params(o.params), "params after 3 steps")
    Xe = rs.uniform(0, 1, size=(50, D))
    code = h.encode(Xe)
    assert code.shape == (50, 6)
    _close(code, o.encode(Xe), "encode")
    _close(h.decode(code), o.feed_forward(Xe), "reconstruction")
    h.close()


@pytest.mark.gpu
def test_ae_estimator_like_reference_test():
    """Shapes the reference test touches (test_autoencoder.py:19-50) and fit/encode/feed_forward parity with the oracle,
    on MNIST-shaped synthetic rows (the reference downloads MNIST; no network here)."""
    from smrt_amd.autoencode import AutoEncoder
    rs = np.random.RandomState(5)
    X = (rs.uniform(size=(700, 784)) * (rs.uniform(size=(700, 784)) > 0.8))
    X_train, X_test = X[:600], X[600:]
    ae = AutoEncoder(n_hidden=400, n_epochs=3, learning_rate=0.01, batch_size=256, display_step=5,
                     activation_function="sigmoid", verbose=0, random_state=42, layer_type="gaussian",
                     early_stopping=True).fit(X_train)
    assert ae.topography_.shape == (784, 400, 784)
    assert ae.transform(X_train).shape == (600, 400)
    rec = ae.feed_forward(X_test)
    assert rec.shape == X_test.shape and rec.dtype == np.float32
    w, b = ae.topography_.get_weights_biases()
    assert [m.shape for m in w] == [(784, 400), (400, 784)] and [v.shape for v in b] == [(400,), (784,)]

    o = OracleAE(784, [400], activation="sigmoid", learning_function="rms_prop", learning_rate=0.01, seed=42,
                 layer_type="gaussian")
    ae2 = AutoEncoder(n_hidden=400, n_epochs=2, learning_rate=0.01, batch_size=256, random_state=42, layer_type="gaussian")
    ae2.set_initial_weights(o.params).fit(X_train)
    o.fit(X_train, n_epochs=2, batch_size=256)
    np.testing.assert_allclose(ae2.epoch_costs_, o.epoch_costs_, rtol=1e-4)
    _close(ae2.encode(X_test), o.encode(X_test), "encode after fit")
    _close(ae2.feed_forward(X_test), o.feed_forward(X_test), "feed_forward after fit")
    with pytest.raises(ValueError):
        ae2.encode(X_test[:, :10])
    ae.clean_session(); ae2.clean_session()
