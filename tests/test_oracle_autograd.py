"""The oracle's hand-derived backward vs torch CPU autograd on the same forward expression.

Independent check named in SURVEY.md section 8(c): validates gradients of
autoencoder.py:469-480 / _ae_utils.py:15-49 / layer.py:279,365 as restated in oracle/.
"""
import numpy as np
import pytest
import torch

from oracle.vae_oracle import ACTIVATIONS, OracleVAE

TORCH_ACT = {
    "sigmoid": torch.sigmoid, "tanh": torch.tanh, "relu": torch.relu,
    "elu": torch.nn.functional.elu, "identity": lambda v: v,
}


def torch_cost(params, X, eps, act, L, lo, hi, l2, masks, keep):
    f = TORCH_ACT[act]
    Ws = [W if masks is None else W / keep * m for (W, _), m in zip(params, masks or [None] * len(params))]
    bs = [b for _, b in params]
    h = X
    for i in range(L):
        h = f(h @ Ws[i] + bs[i])
    mu = f(h @ Ws[L] + bs[L])
    ls = f(h @ Ws[L + 1] + bs[L + 1])
    z = mu + eps * torch.exp(ls)
    d = z
    for j in range(L + 1):
        idx = L + 2 + j
        pre = d @ Ws[idx] + bs[idx]
        d = torch.sigmoid(pre) if j == L else f(pre)
    pc = torch.clamp(d, lo, hi)
    rec = -(X * torch.log(pc) + (1 - X) * torch.log(1 - pc)).sum(1)
    kl = -0.5 * (1 + 2 * ls - mu ** 2 - torch.exp(2 * ls)).sum(1)
    c = (rec + kl).mean()
    if l2 is not None:
        reg = list(range(L)) + list(range(L + 2, 2 * L + 3))
        c = c + l2 * sum(0.5 * (Ws[i] ** 2).sum() for i in reg)
    return c


@pytest.mark.parametrize("act", ACTIVATIONS)
@pytest.mark.parametrize("hidden", [[7], [9, 6]])
@pytest.mark.parametrize("l2,drop", [(1e-4, 1.0), (None, 1.0), (1e-2, 0.5)])
def test_backward_matches_autograd(act, hidden, l2, drop):
    rs = np.random.RandomState(0)
    D, Z, B = 12, 3, 5
    X = rs.uniform(0, 1, size=(B, D))
    eps = rs.normal(size=(B, Z))
    o = OracleVAE(D, hidden, Z, activation=act, l2_penalty=l2, dropout=drop, dtype=np.float64,
                  layer_type="gaussian", seed=3)
    # decorrelate same-shaped layers (init gives identical heads on purpose)
    o.params = [(W + 0.1 * rs.normal(size=W.shape), b + 0.1 * rs.normal(size=b.shape)) for W, b in o.params]
    masks = None
    if drop < 1.0:
        masks = [np.floor(drop + rs.uniform(size=W.shape)) for W, _ in o.params]
    cache = o.forward(X, eps, masks)
    c = o.cost(cache)
    gW, gb = o.gradients(cache, masks)

    tp = [(torch.tensor(W, requires_grad=True), torch.tensor(b, requires_grad=True)) for W, b in o.params]
    tm = None if masks is None else [torch.tensor(m) for m in masks]
    tc = torch_cost(tp, torch.tensor(X), torch.tensor(eps), act, len(hidden), float(o.lo), float(o.hi),
                    None if l2 is None else float(l2), tm, float(drop))
    tc.backward()
    assert np.isclose(float(tc), float(c), rtol=1e-12)
    for (W, b), g_w, g_b in zip(tp, gW, gb):
        np.testing.assert_allclose(g_w, W.grad.numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g_b, b.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_clip_gradient_gate_matches_autograd():
    """[TF-sem] gradient flows through clip_by_value only where lo <= p <= hi; torch.clamp agrees."""
    rs = np.random.RandomState(1)
    D, Z, B = 6, 2, 4
    X = rs.uniform(0, 1, size=(B, D))
    eps = rs.normal(size=(B, Z))
    o = OracleVAE(D, [5], Z, eps=0.2, dtype=np.float64, l2_penalty=None)   # wide clip so it is active
    o.params = [(W + rs.normal(size=W.shape), b + rs.normal(size=b.shape)) for W, b in o.params]
    cache = o.forward(X, eps)
    p = cache["dec"][-1]
    assert ((p < 0.2) | (p > 0.8)).any() and ((p > 0.2) & (p < 0.8)).any()
    gW, gb = o.gradients(cache)
    tp = [(torch.tensor(W, requires_grad=True), torch.tensor(b, requires_grad=True)) for W, b in o.params]
    tc = torch_cost(tp, torch.tensor(X), torch.tensor(eps), "sigmoid", 1, 0.2, 0.8, None, None, 1.0)
    tc.backward()
    for (W, b), g_w, g_b in zip(tp, gW, gb):
        np.testing.assert_allclose(g_w, W.grad.numpy(), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(g_b, b.grad.numpy(), rtol=1e-9, atol=1e-12)
