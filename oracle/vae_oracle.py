"""numpy restatement of the reference's VAE arithmetic (fp32 or fp64).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED against
TensorFlow (absent third-party dependency, tensorflow>=1.0, CI pin 1.0.0:
requirements.txt:4, build_tools/travis/install.sh:51,54): this file restates the
published TF-1.x semantics at every reference call site and is anchored on those
call sites.  Items tagged [TF-sem] are TF behaviour, everything else is read off
the reference source directly.

All citations are relative to /root/reference/.

Layer order used everywhere ("ABI order"):
    enc_1 .. enc_L, z_mean head, z_log_sigma head, dec_1 .. dec_{L+1}
with W stored [fan_in, fan_out] row-major and b [fan_out]
(smrt/autoencode/layer.py:292-340).
"""
from __future__ import annotations

import time

import numpy as np

ACTIVATIONS = ("elu", "identity", "relu", "sigmoid", "tanh")       # autoencoder.py:28-34
# autoencoder.py:38-48.  'momentum' and 'adagrad-da' cannot be constructed the way
# the reference calls them (learning_function(rate), base.py:138,141): TypeError.
OPTIMIZERS = ("adadelta", "adagrad", "adam", "proximal-sgd", "proximal-adagrad", "rms_prop", "sgd")
BROKEN_OPTIMIZERS = ("momentum", "adagrad-da")


# --------------------------------------------------------------------------- topology
def resolve_n_latent(n_latent_factors, n_features):
    """layer.py:249-259 -- int passes through, float f in (0,1) -> max(2, round(f*D))."""
    if isinstance(n_latent_factors, (bool, np.bool_)):
        raise ValueError("n_latent_factors must be an int >= 0")
    if isinstance(n_latent_factors, (int, np.integer)):
        if int(n_latent_factors) < 0:
            raise ValueError("n_latent_factors must be an int >= 0")
        return int(n_latent_factors)
    if isinstance(n_latent_factors, (float, np.floating)):
        f = float(n_latent_factors)
        if not (0.0 < f < 1.0):
            raise ValueError("Expected 0 < n_latent_factors < 1.0")
        return max(2, int(round(f * n_features)))
    raise TypeError("n_latent_factors must be an int or a float")


def layer_dims(n_features, n_hidden, n_latent):
    """(fan_in, fan_out) of every dense layer in ABI order -- layer.py:292-297,312-317."""
    hidden = [int(n_hidden)] if not isinstance(n_hidden, (list, tuple)) else [int(h) for h in n_hidden]
    widths = [int(n_features)] + hidden
    enc = list(zip(widths[:-1], widths[1:]))
    dec = [(n_latent, widths[-1])] + [(b, a) for a, b in reversed(enc)]
    heads = [(widths[-1], n_latent), (widths[-1], n_latent)]
    return enc, heads, dec


def topography_shape(n_features, n_hidden, n_latent):
    """layer.py:60-72 -- encoder fan-ins, decoder fan-ins, output width."""
    enc, _, dec = layer_dims(n_features, n_hidden, n_latent)
    return tuple([fi for fi, _ in enc] + [fi for fi, _ in dec] + [dec[-1][1]])


def init_params(n_features, n_hidden, n_latent, layer_type="xavier", bias_strategy="zeros",
                seed=42, dtype=np.float32):
    """Initial weights.

    The reference draws them from TensorFlow's Philox stream (layer.py:414,460), which
    cannot be reproduced without TensorFlow; the *distributions* are restated
    (Xavier U(+-sqrt(6/(fi+fo))) layer.py:457-461; Gaussian N(0, sqrt(2/fi)) layer.py:412-414;
    bias zeros/ones layer.py:472-475) on a documented numpy stream.  [TF-sem] every layer
    is seeded with the same (graph, op) seed pair (layer.py:165,285 + base.py:171), so
    same-shaped layers start identical -- kept here by restarting RandomState(seed) per layer.
    """
    if layer_type not in ("xavier", "gaussian"):
        raise ValueError("layer_type must be one of ['gaussian', 'xavier']")
    if bias_strategy not in ("zeros", "ones"):
        raise ValueError("bias_strategy must be one of ['ones', 'zeros']")
    enc, heads, dec = layer_dims(n_features, n_hidden, n_latent)
    params = []
    for fi, fo in enc + heads + dec:
        rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        if layer_type == "xavier":
            lim = np.sqrt(6.0 / (fi + fo))
            W = rs.uniform(-lim, lim, size=(fi, fo))
        else:
            W = rs.normal(0.0, (2.0 / fi) ** 0.5, size=(fi, fo))
        b = np.zeros(fo) if bias_strategy == "zeros" else np.ones(fo)
        params.append((W.astype(dtype), b.astype(dtype)))
    return params


def flatten_params(params):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in params])


def unflatten_params(flat, n_features, n_hidden, n_latent, dtype=None):
    enc, heads, dec = layer_dims(n_features, n_hidden, n_latent)
    out, o = [], 0
    for fi, fo in enc + heads + dec:
        W = np.array(flat[o:o + fi * fo], dtype=dtype).reshape(fi, fo); o += fi * fo
        b = np.array(flat[o:o + fo], dtype=dtype); o += fo
        out.append((W, b))
    assert o == len(flat)
    return out


# --------------------------------------------------------------------------- activations
def act_forward(name, v):
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-v))                      # [TF-sem] Eigen scalar_sigmoid_op
    if name == "tanh":
        return np.tanh(v)
    if name == "relu":
        return np.maximum(v, 0)
    if name == "elu":
        return np.where(v > 0, v, np.expm1(np.minimum(v, 0)))   # [TF-sem] exp(x)-1 for x<0
    if name == "identity":
        return v
    raise ValueError("Permitted activation functions: %r" % list(ACTIVATIONS))


def act_grad_from_output(name, y, dy):
    """[TF-sem] every TF-1.x activation gradient is a function of the OUTPUT."""
    if name == "sigmoid":
        return dy * y * (1 - y)
    if name == "tanh":
        return dy * (1 - y * y)
    if name == "relu":
        return dy * (y > 0)
    if name == "elu":
        return np.where(y < 0, dy * (y + 1), dy)
    if name == "identity":
        return dy
    raise ValueError(name)


# --------------------------------------------------------------------------- Philox (own stream)
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
STREAM_EPS, STREAM_DROPOUT, STREAM_GENERATE = 1, 2, 3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counter-based Philox4x32-10 (Salmon et al. 2011); vectorised over counters."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & np.uint64(0xFFFFFFFF) for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def _box_muller_f32(xa, xb):
    f = np.float32
    u1 = ((xa >> np.uint32(8)).astype(f) + f(1.0)) * f(2.0 ** -24)        # (0, 1]
    u2 = (xb >> np.uint32(8)).astype(f) * f(2.0 ** -24)                   # [0, 1)
    r = np.sqrt(f(-2.0) * np.log(u1)).astype(f)
    # the angle 2 pi u2 in fp64, cos / sin rounded to fp32 once: the kernel reduces the quadrant exactly (in revolutions)
    # and evaluates a ~1 ulp polynomial, so both sides are within an ulp or two of the exact value
    th = 6.283185307179586 * u2.astype(np.float64)
    return (r * np.cos(th).astype(f)).astype(f), (r * np.sin(th).astype(f)).astype(f)


def philox_normal(seed, stream, step, first_elem, n_elems):
    """n_elems standard normals (fp32) for flat element indices first_elem.. of a stream.

    Element e lives in Philox block q = e // 4 with counter
    (q & 0xffffffff, q >> 32, step, stream) and key (seed lo, seed hi); outputs
    (x0,x1)->(n0,n1), (x2,x3)->(n2,n3) by Box-Muller.  This is the kernel's stream
    (smrt_amd/csrc/vae_kernels.hip: philox_normal4), not TensorFlow's.
    """
    e = np.arange(first_elem, first_elem + n_elems, dtype=np.uint64)
    q = np.unique(e // np.uint64(4))
    x = philox4x32_10(q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.uint64(step), np.uint64(stream),
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    n0, n1 = _box_muller_f32(x[0], x[1])
    n2, n3 = _box_muller_f32(x[2], x[3])
    allv = np.stack([n0, n1, n2, n3], axis=1).reshape(-1)
    off = int(first_elem - int(q[0]) * 4)
    return allv[off:off + n_elems]


def philox_uniform(seed, stream, step, first_elem, n_elems):
    e0 = int(first_elem)
    q = np.arange(e0 // 4, (e0 + n_elems + 3) // 4 + 1, dtype=np.uint64)
    x = philox4x32_10(q & np.uint64(0xFFFFFFFF), q >> np.uint64(32), np.uint64(step), np.uint64(stream),
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    u = np.stack(x, axis=1).reshape(-1)
    u = (u >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    off = e0 - int(q[0]) * 4
    return u[off:off + n_elems]


# --------------------------------------------------------------------------- optimizers [TF-sem]
class _Opt(object):
    def __init__(self, name, lr, shapes, dtype):
        if name in BROKEN_OPTIMIZERS:
            # tf.train.MomentumOptimizer / AdagradDAOptimizer need a 2nd positional argument;
            # the reference calls learning_function(rate) (base.py:138,141) -> TypeError.
            raise TypeError("__init__() missing 1 required positional argument (learning_function=%r)" % name)
        if name not in OPTIMIZERS:
            raise ValueError("Permitted learning functions: %r" % sorted(OPTIMIZERS + BROKEN_OPTIMIZERS))
        self.name, self.dt = name, dtype
        self.lr = dtype(lr)
        z = lambda v: [np.full(s, v, dtype=dtype) for s in shapes]
        if name == "rms_prop":
            self.ms, self.mom = z(1.0), z(0.0)                    # [TF-sem] rms slot starts at ONE
        elif name == "adam":
            self.m, self.v = z(0.0), z(0.0)
            self.b1p, self.b2p = dtype(0.9), dtype(0.999)         # beta power accumulators
        elif name in ("adagrad", "proximal-adagrad"):
            self.acc = z(0.1)                                     # initial_accumulator_value
        elif name == "adadelta":
            self.acc, self.acc_upd = z(0.0), z(0.0)

    def state_arrays(self):
        """Slot arrays as a dict name -> list (per variable), for parity checks."""
        return {k: v for k, v in self.__dict__.items() if isinstance(v, list)}

    def apply(self, variables, grads):
        dt, lr = self.dt, self.lr
        n = self.name
        if n == "adam":
            b1, b2, eps = dt(0.9), dt(0.999), dt(1e-8)
            alpha = lr * np.sqrt(dt(1) - self.b2p) / (dt(1) - self.b1p)
        for i, (w, g) in enumerate(zip(variables, grads)):
            if n in ("sgd", "proximal-sgd"):                       # prox with l1=l2=0 == sgd
                w -= lr * g
            elif n == "rms_prop":                                  # decay .9, momentum 0, eps 1e-10
                ms, mom = self.ms[i], self.mom[i]
                ms += (g * g - ms) * (dt(1) - dt(0.9))
                mom[...] = mom * dt(0.0) + lr * g / np.sqrt(ms + dt(1e-10))
                w -= mom
            elif n == "adam":
                m, v = self.m[i], self.v[i]
                m += (g - m) * (dt(1) - b1)
                v += (g * g - v) * (dt(1) - b2)
                w -= alpha * m / (np.sqrt(v) + eps)
            elif n in ("adagrad", "proximal-adagrad"):
                acc = self.acc[i]
                acc += g * g
                w -= lr * g / np.sqrt(acc)
            elif n == "adadelta":                                  # rho .95, eps 1e-8
                acc, au = self.acc[i], self.acc_upd[i]
                rho, eps = dt(0.95), dt(1e-8)
                acc[...] = acc * rho + g * g * (dt(1) - rho)
                upd = np.sqrt(au + eps) / np.sqrt(acc + eps) * g
                au[...] = au * rho + upd * upd * (dt(1) - rho)
                w -= lr * upd
        if n == "adam":
            self.b1p = dt(self.b1p * dt(0.9))
            self.b2p = dt(self.b2p * dt(0.999))


# --------------------------------------------------------------------------- the model
class OracleVAE(object):
    """Restatement of VariationalAutoEncoder's graph + training loop.

    Follows smrt/autoencode/autoencoder.py:439-486 (graph), layer.py:272-345 (topography),
    _ae_utils.py:15-49 (losses), base.py:127-149 (L2, clip), base.py:183-254 (loop).
    """

    def __init__(self, n_features, n_hidden, n_latent, activation="sigmoid", learning_function="rms_prop",
                 learning_rate=0.05, l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0,
                 eps=1e-10, dropout=1.0, seed=42, dtype=np.float32, params=None,
                 layer_type="xavier", bias_strategy="zeros"):
        if activation not in ACTIVATIONS:
            raise ValueError("Permitted activation functions: %r" % list(ACTIVATIONS))
        self.dt = np.dtype(dtype).type
        self.D, self.Z = int(n_features), int(n_latent)
        self.hidden = [int(n_hidden)] if not isinstance(n_hidden, (list, tuple)) else [int(h) for h in n_hidden]
        self.L = len(self.hidden)
        self.act = activation
        self.l2 = None if l2_penalty is None else self.dt(l2_penalty)
        self.clip, self.gmin, self.gmax = bool(clip), self.dt(gclip_min), self.dt(gclip_max)
        # _ae_utils.py:31 -- clip bounds become constants of the tensor dtype [TF-sem]:
        # float32(1 - 1e-10) == 1.0f, i.e. no effective upper clip in fp32.
        self.lo, self.hi = self.dt(eps), self.dt(1.0 - eps)
        self.keep = self.dt(dropout)
        self.seed = int(seed)
        if params is None:
            params = init_params(n_features, self.hidden, n_latent, layer_type, bias_strategy, seed, dtype)
        self.params = [(np.array(W, dtype=dtype), np.array(b, dtype=dtype)) for W, b in params]
        shapes = []
        for W, b in self.params:
            shapes += [W.shape, b.shape]
        self.opt = _Opt(learning_function, learning_rate, shapes, self.dt)
        self.step = 0

    # ---- forward (layer.py:299-340, 348-368)
    def _effective_weights(self, masks):
        """tf.nn.dropout on the WEIGHT matrix (layer.py:365): W' = W / keep * mask [TF-sem]."""
        if masks is None:
            return [W for W, _ in self.params]
        return [(W / self.keep) * m.astype(W.dtype) for (W, _), m in zip(self.params, masks)]

    def forward(self, X, eps, masks=None):
        dt, L = self.dt, self.L
        X = np.asarray(X).astype(dt)                       # fp64 -> fp32 feed cast (base.py:174)
        Ws = self._effective_weights(masks)
        bs = [b for _, b in self.params]
        acts = [X]
        for i in range(L):
            acts.append(act_forward(self.act, acts[-1] @ Ws[i] + bs[i]).astype(dt))
        h = acts[-1]
        mu = act_forward(self.act, h @ Ws[L] + bs[L]).astype(dt)          # heads use the user
        ls = act_forward(self.act, h @ Ws[L + 1] + bs[L + 1]).astype(dt)  # activation, layer.py:313
        cache = dict(X=X, enc=acts, mu=mu, ls=ls, Ws=Ws)
        if eps is None:
            return cache
        eps = np.asarray(eps).astype(dt)
        z = (mu + eps * np.exp(ls)).astype(dt)                            # layer.py:279
        cache.update(eps=eps, z=z)
        cache["dec"] = self._decode_chain(z, Ws, bs)
        return cache

    def _decode_chain(self, z, Ws, bs):
        L, dt = self.L, self.dt
        outs = [z]
        n_dec = L + 1
        for j in range(n_dec):
            idx = L + 2 + j
            name = "sigmoid" if j == n_dec - 1 else self.act               # layer.py:333-338
            outs.append(act_forward(name, outs[-1] @ Ws[idx] + bs[idx]).astype(dt))
        return outs

    # ---- loss (autoencoder.py:469-480, _ae_utils.py:15-49, base.py:127-133)
    def cost(self, cache):
        dt = self.dt
        X, p, mu, ls = cache["X"], cache["dec"][-1], cache["mu"], cache["ls"]
        pc = np.minimum(np.maximum(p, self.lo), self.hi)
        with np.errstate(divide="ignore", invalid="ignore"):
            rec = -np.sum(X * np.log(pc) + (dt(1) - X) * np.log(dt(1) - pc), axis=1)
        kl = dt(-0.5) * np.sum(dt(1) + dt(2) * ls - mu * mu - np.exp(dt(2) * ls), axis=1)
        c = np.mean(rec + kl)
        if self.l2 is not None:
            L = self.L
            reg_idx = list(range(L)) + list(range(L + 2, 2 * L + 3))        # enc + dec, NOT the heads (layer.py:96)
            c = c + self.l2 * sum(dt(0.5) * np.sum(cache["Ws"][i] ** 2) for i in reg_idx)
        return dt(c)

    # ---- backward: hand-derived, checked against torch autograd in tests/
    def gradients(self, cache, masks=None, inv_batch=None):
        dt, L = self.dt, self.L
        X, mu, ls, eps, Ws = cache["X"], cache["mu"], cache["ls"], cache["eps"], cache["Ws"]
        enc, dec = cache["enc"], cache["dec"]
        B = X.shape[0]
        invB = dt(1.0 / B) if inv_batch is None else dt(inv_batch)
        nl = len(self.params)
        gW, gb = [None] * nl, [None] * nl
        p = dec[-1]
        pc = np.minimum(np.maximum(p, self.lo), self.hi)
        with np.errstate(divide="ignore", invalid="ignore"):
            dpc = -(X / pc - (dt(1) - X) / (dt(1) - pc)) * invB
        # [TF-sem] clip_by_value = minimum(maximum(p, lo), hi); Maximum passes grad where
        # p >= lo, Minimum where max(p, lo) <= hi.
        passes = (p >= self.lo) & (np.maximum(p, self.lo) <= self.hi)
        dp = np.where(passes, dpc, dt(0))
        dpre = act_grad_from_output("sigmoid", p, dp).astype(dt)
        n_dec = L + 1
        for j in reversed(range(n_dec)):
            idx = L + 2 + j
            a_in = dec[j]
            gW[idx] = (a_in.T @ dpre).astype(dt)
            gb[idx] = dpre.sum(axis=0).astype(dt)
            d_in = (dpre @ Ws[idx].T).astype(dt)
            if j > 0:
                dpre = act_grad_from_output(self.act, a_in, d_in).astype(dt)
        dz = d_in
        # reparameterisation + KL (layer.py:279, _ae_utils.py:49)
        dmu = dz + mu * invB
        dls = dz * eps * np.exp(ls) + (np.exp(dt(2) * ls) - dt(1)) * invB
        dpre_mu = act_grad_from_output(self.act, mu, dmu).astype(dt)
        dpre_ls = act_grad_from_output(self.act, ls, dls).astype(dt)
        h = enc[-1]
        gW[L], gb[L] = (h.T @ dpre_mu).astype(dt), dpre_mu.sum(axis=0).astype(dt)
        gW[L + 1], gb[L + 1] = (h.T @ dpre_ls).astype(dt), dpre_ls.sum(axis=0).astype(dt)
        dh = (dpre_mu @ Ws[L].T + dpre_ls @ Ws[L + 1].T).astype(dt)
        for i in reversed(range(L)):
            dpre = act_grad_from_output(self.act, enc[i + 1], dh).astype(dt)
            gW[i] = (enc[i].T @ dpre).astype(dt)
            gb[i] = dpre.sum(axis=0).astype(dt)
            if i > 0:
                dh = (dpre @ Ws[i].T).astype(dt)
        # L2 on the post-dropout tensors of enc+dec layers (base.py:129, layer.py:96,365)
        if self.l2 is not None:
            for i in list(range(L)) + list(range(L + 2, 2 * L + 3)):
                gW[i] = (gW[i] + self.l2 * Ws[i]).astype(dt)
        # back through dropout to the variable: dW = dW' * mask / keep [TF-sem]
        if masks is not None:
            gW = [(g / self.keep) * m.astype(dt) for g, m in zip(gW, masks)]
        return gW, gb

    def clip_grads(self, gW, gb):
        """base.py:143-147 -- element-wise value clip of every trainable variable's gradient."""
        if not self.clip:
            return gW, gb
        c = lambda g: np.minimum(np.maximum(g, self.gmin), self.gmax)
        return [c(g) for g in gW], [c(g) for g in gb]

    # ---- one sess.run([optimizer, cost]) (base.py:215-217)
    def draw_eps(self, rows, row0=0):
        """Kernel-stream epsilon for the current step (used when nothing is injected)."""
        return philox_normal(self.seed, STREAM_EPS, self.step, row0 * self.Z, rows * self.Z).reshape(rows, self.Z)

    def draw_masks(self):
        """Kernel-stream dropout masks for the current step: floor(keep + U[0,1)) [TF-sem]."""
        if float(self.keep) >= 1.0:
            return None
        masks, off = [], 0
        for W, _ in self.params:
            u = philox_uniform(self.seed, STREAM_DROPOUT, self.step, off, W.size).reshape(W.shape)
            masks.append(np.floor(self.keep + u).astype(self.dt))
            off += W.size
        return masks

    def train_step(self, X, eps=None, masks=None, return_grads=False):
        if eps is None:
            eps = self.draw_eps(len(X))
        if masks is None:
            masks = self.draw_masks()
        cache = self.forward(X, eps, masks)
        c = self.cost(cache)
        gW, gb = self.gradients(cache, masks)
        cW, cb = self.clip_grads(gW, gb)
        variables, grads = [], []
        for (W, b), g_w, g_b in zip(self.params, cW, cb):
            variables += [W, b]
            grads += [g_w, g_b]
        self.opt.apply(variables, grads)
        self.step += 1
        if return_grads:
            return c, (gW, gb)
        return c

    # ---- the training loop (base.py:183-254)
    def fit(self, X, n_epochs=20, batch_size=128, early_stopping=False, min_change=1e-3, eps_fn=None):
        X = np.asarray(X, dtype=np.float64)
        n = X.shape[0]
        starts = list(range(0, n, batch_size))                    # sklearn gen_batches: contiguous, ragged tail
        epoch_times, costs, last_cost, c = [], [], None, None
        for epoch in range(n_epochs):
            t0 = time.time()
            for s in starts:
                chunk = X[s:s + batch_size]
                c = self.train_step(chunk, None if eps_fn is None else eps_fn(self.step, len(chunk)))
            epoch_times.append(time.time() - t0)
            costs.append(c)                                       # cost of the LAST batch only (base.py:222)
            if early_stopping:
                if last_cost is None:
                    last_cost = c
                elif abs(last_cost - c) <= min_change:
                    break
                else:
                    last_cost = c
        self.train_cost_, self.epoch_times_, self.epoch_costs_ = c, epoch_times, costs
        return self

    # ---- inference (autoencoder.py:488-572)
    def encode(self, X):
        cache = self.forward(X, None)
        return [cache["mu"], cache["ls"]]                        # list of two fp32 arrays (autoencoder.py:500)

    def decode(self, Z):
        Z = np.asarray(Z, dtype=np.float64).astype(self.dt)      # check_array fp64, fed to the fp32 tensor
        Ws = [W for W, _ in self.params]
        bs = [b for _, b in self.params]
        return self._decode_chain(Z, Ws, bs)[-1]

    def generate(self, random_state, n=1, **nrm_args):
        z = random_state.normal(size=(n, self.Z), **nrm_args)    # autoencoder.py:571
        return self.decode(z)

    def generate_from_sample(self, random_state, X, **nrm_args):
        z_mu, log_sigma = self.encode(np.asarray(X, dtype=np.float64))
        z_mu = z_mu.copy()
        eps = random_state.normal(size=log_sigma.shape, **nrm_args)   # autoencoder.py:544
        z_mu += eps * np.exp(log_sigma)                               # fp64 maths, stored in place (:545)
        return self.decode(z_mu)
