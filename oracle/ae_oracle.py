"""numpy restatement of the reference's plain ``AutoEncoder`` arithmetic (fp32 or fp64).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED against TensorFlow, for the same
reason as ``vae_oracle.py`` (TF is an absent third-party dependency and the reference's AE test pins no number:
smrt/autoencode/tests/test_autoencoder.py:39,50 are commented out).  Citations are relative to /root/reference/.

Layer order ("ABI order"): enc_1 .. enc_L, dec_1 .. dec_L (smrt/autoencode/layer.py:163-190), W [fan_in, fan_out].
"""
from __future__ import annotations

import time

import numpy as np

from .vae_oracle import ACTIVATIONS, _Opt, act_forward, act_grad_from_output


def ae_layer_dims(n_features, n_hidden):
    """layer.py:163-166 -- encoder pyramid and its mirror."""
    hidden = [int(n_hidden)] if not isinstance(n_hidden, (list, tuple)) else [int(h) for h in n_hidden]
    widths = [int(n_features)] + hidden
    enc = list(zip(widths[:-1], widths[1:]))
    dec = [(b, a) for a, b in reversed(enc)]
    return enc, dec


def ae_init_params(n_features, n_hidden, layer_type="xavier", bias_strategy="zeros", seed=42, dtype=np.float32):
    """Same distributions and the same per-layer restart of the stream as vae_oracle.init_params."""
    enc, dec = ae_layer_dims(n_features, n_hidden)
    params = []
    for fi, fo in enc + dec:
        rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        if layer_type == "xavier":
            lim = np.sqrt(6.0 / (fi + fo))
            W = rs.uniform(-lim, lim, size=(fi, fo))
        elif layer_type == "gaussian":
            W = rs.normal(0.0, (2.0 / fi) ** 0.5, size=(fi, fo))
        else:
            raise ValueError("layer_type must be one of ['gaussian', 'xavier']")
        b = np.zeros(fo) if bias_strategy == "zeros" else np.ones(fo)
        params.append((W.astype(dtype), b.astype(dtype)))
    return params


class OracleAE(object):
    """AutoEncoder graph + loop: autoencoder.py:234-282 (graph, encode, feed_forward), base.py:127-149,183-254."""

    def __init__(self, n_features, n_hidden, activation="sigmoid", learning_function="rms_prop", learning_rate=0.05,
                 l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, seed=42, dtype=np.float32, params=None,
                 layer_type="xavier", bias_strategy="zeros"):
        if activation not in ACTIVATIONS:
            raise ValueError("Permitted activation functions: %r" % list(ACTIVATIONS))
        self.dt = np.dtype(dtype).type
        self.D = int(n_features)
        self.hidden = [int(n_hidden)] if not isinstance(n_hidden, (list, tuple)) else [int(h) for h in n_hidden]
        self.L = len(self.hidden)
        self.act = activation
        self.l2 = None if l2_penalty is None else self.dt(l2_penalty)
        self.clip, self.gmin, self.gmax = bool(clip), self.dt(gclip_min), self.dt(gclip_max)
        if params is None:
            params = ae_init_params(n_features, self.hidden, layer_type, bias_strategy, seed, dtype)
        self.params = [(np.array(W, dtype=dtype), np.array(b, dtype=dtype)) for W, b in params]
        shapes = []
        for W, b in self.params:
            shapes += [W.shape, b.shape]
        self.opt = _Opt(learning_function, learning_rate, shapes, self.dt)
        self.step = 0

    def forward(self, X):
        """Every layer, the output layer included, applies the user's activation (layer.py:171-177, 348-368)."""
        X = np.asarray(X).astype(self.dt)
        outs = [X]
        for W, b in self.params:
            outs.append(act_forward(self.act, outs[-1] @ W + b).astype(self.dt))
        return outs

    def cost(self, outs):
        """reduce_mean(pow(y_true - y_pred, 2)) over ALL elements + l2 * sum 1/2 |W|^2 (autoencoder.py:260, base.py:127-133)."""
        dt = self.dt
        d = outs[0] - outs[-1]
        c = np.mean(d * d)
        if self.l2 is not None:
            c = c + self.l2 * sum(dt(0.5) * np.sum(W ** 2) for W, _ in self.params)
        return dt(c)

    def gradients(self, outs):
        dt = self.dt
        B, D = outs[0].shape
        dy = (dt(2.0) * (outs[-1] - outs[0]) * dt(1.0 / (B * D))).astype(dt)
        gW, gb = [None] * len(self.params), [None] * len(self.params)
        for i in reversed(range(len(self.params))):
            dpre = act_grad_from_output(self.act, outs[i + 1], dy).astype(dt)
            gW[i] = (outs[i].T @ dpre).astype(dt)
            gb[i] = dpre.sum(axis=0).astype(dt)
            if i > 0:
                dy = (dpre @ self.params[i][0].T).astype(dt)
        if self.l2 is not None:
            gW = [(g + self.l2 * W).astype(dt) for g, (W, _) in zip(gW, self.params)]
        return gW, gb

    def train_step(self, X, return_grads=False):
        outs = self.forward(X)
        c = self.cost(outs)
        gW, gb = self.gradients(outs)
        cW, cb = gW, gb
        if self.clip:                                            # base.py:143-147
            f = lambda g: np.minimum(np.maximum(g, self.gmin), self.gmax)
            cW, cb = [f(g) for g in gW], [f(g) for g in gb]
        variables, grads = [], []
        for (W, b), a, d in zip(self.params, cW, cb):
            variables += [W, b]
            grads += [a, d]
        self.opt.apply(variables, grads)
        self.step += 1
        return (c, (gW, gb)) if return_grads else c

    def fit(self, X, n_epochs=20, batch_size=128, early_stopping=False, min_change=1e-3):
        X = np.asarray(X, dtype=np.float64)
        costs, times, last, c = [], [], None, None
        for _ in range(n_epochs):
            t0 = time.time()
            for s in range(0, X.shape[0], batch_size):
                c = self.train_step(X[s:s + batch_size])
            times.append(time.time() - t0)
            costs.append(c)                                      # last batch only (base.py:222)
            if early_stopping:
                if last is None:
                    last = c
                elif abs(last - c) <= min_change:
                    break
                else:
                    last = c
        self.train_cost_, self.epoch_times_, self.epoch_costs_ = c, times, costs
        return self

    def encode(self, X):
        return self.forward(X)[self.L]                           # autoencoder.py:262-266

    def decode(self, code):
        out = np.asarray(code, dtype=np.float64).astype(self.dt)
        for W, b in self.params[self.L:]:
            out = act_forward(self.act, out @ W + b).astype(self.dt)
        return out

    def feed_forward(self, X):
        return self.forward(X)[-1]                               # autoencoder.py:268-282
