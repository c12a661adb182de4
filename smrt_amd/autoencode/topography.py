"""Network wiring of the symmetrical VAE and host-side weight initialisation.

Host mirror of the reference's ``SymmetricalVAETopography`` / dense layer classes
(smrt/autoencode/layer.py:193-345, 348-464): it carries dimensions, the ``shape`` tuple and
numpy views of the weights; the arithmetic itself lives in the HIP library.
"""
from __future__ import annotations

import threading

import numpy as np

from ..utils import get_random_state

__all__ = ["VAETopography", "DenseLayerView", "resolve_latent", "normalize_hidden", "initial_parameters",
           "PERMITTED_LAYER_TYPES", "PERMITTED_BIAS_STRATEGIES"]

PERMITTED_LAYER_TYPES = ("gaussian", "xavier")        # layer.py:467-470
PERMITTED_BIAS_STRATEGIES = ("ones", "zeros")         # layer.py:472-475


def normalize_hidden(n_hidden):
    """int -> [int]; list kept (a COPY: the reference inserts D into the caller's list, layer.py:292)."""
    if isinstance(n_hidden, list):
        return [int(v) for v in n_hidden]
    if isinstance(n_hidden, (bool, np.bool_)) or not isinstance(n_hidden, (int, np.integer)):
        raise ValueError("n_hidden must be an int or list")
    return [int(n_hidden)]


def resolve_latent(n_latent_factors, n_features):
    """int >= 0 passes; float f in (0,1) compresses to max(2, round(f * D)); else TypeError (layer.py:249-259)."""
    if isinstance(n_latent_factors, (bool, np.bool_)):
        raise ValueError("n_latent_factors must be an int >= 0")
    if isinstance(n_latent_factors, (int, np.integer)):
        if int(n_latent_factors) < 0:
            raise ValueError("n_latent_factors must be an int >= 0")
        return int(n_latent_factors)
    if isinstance(n_latent_factors, (float, np.floating)):
        f = float(n_latent_factors)
        if not (0.0 < f < 1.0):
            raise ValueError("Expected 0 < n_latent_factors < 1.0, but got %r" % f)
        return max(2, int(round(f * n_features)))
    raise TypeError("n_latent_factors must be an int or a float")


_STREAM_LOCK = threading.Lock()
_STREAM_CACHE = {}                    # (seed, layer_type) -> the longest unit stream drawn so far (one entry, <= 64 MB)
_STREAM_CACHE_BYTES = 64 << 20


def _unit_stream(seed, layer_type, n):
    """First ``n`` values of ``RandomState(seed).random_sample`` (xavier) or ``.standard_normal`` (gaussian).

    Every layer restarts the generator from the same seed (see ``initial_parameters``), so a layer's draws are a PREFIX
    of the longest layer's, and ``uniform(-b, b)`` / ``normal(0, s)`` are ``-b + (2 b) u`` / ``0 + s g`` of those unit
    draws in numpy's legacy generator (``loc + scale * x``, one multiply and one add in fp64).  Drawing the stream once
    per net instead of once per layer -- and once for all the per-class nets of one ``smrt_balance`` call -- gives the same
    bits (``tests/test_host_goldens.py``) for a seventh of the Mersenne-Twister work on a 4 x 1024 net."""
    key = (int(seed) & 0xFFFFFFFF, layer_type)
    with _STREAM_LOCK:
        u = _STREAM_CACHE.get(key)
        if u is not None and u.shape[0] >= n:
            return u[:n]
    rs = np.random.RandomState(key[0])
    u = rs.random_sample(n) if layer_type == "xavier" else rs.standard_normal(n)
    if u.nbytes <= _STREAM_CACHE_BYTES:
        with _STREAM_LOCK:
            _STREAM_CACHE.clear()
            _STREAM_CACHE[key] = u
    return u


def initial_parameters(dims, layer_type, bias_strategy, seed):
    """Initial (W, b) per layer, fp32, ABI order.

    Xavier: U(+-sqrt(6/(fi+fo))) (layer.py:457-461); Gaussian: N(0, sqrt(2/fi)) (layer.py:412-414).
    The reference draws from TensorFlow's Philox stream with the SAME (graph, op) seed pair for every
    layer, so same-shaped layers (the two latent heads) start identical; that property is kept by
    restarting ``RandomState(seed)`` per layer.  The values themselves are this package's own stream:
    ``RandomState(seed).uniform(-bound, bound, (fi, fo))`` resp. ``.normal(0, sqrt(2/fi), (fi, fo))``, taken here
    from one shared unit stream (``_unit_stream``).
    """
    dims = [(int(fi), int(fo)) for fi, fo in dims]
    u = _unit_stream(seed, layer_type, max([fi * fo for fi, fo in dims] + [0]))
    out, by_shape = [], {}
    for fi, fo in dims:
        W32 = by_shape.get((fi, fo))                                  # same shape = same draws, same scale: same values
        if W32 is None:
            W = np.empty(fi * fo)
            if layer_type == "xavier":
                bound = np.sqrt(6.0 / (fi + fo))
                np.multiply(u[:fi * fo], bound - (-bound), out=W)     # == rs.uniform(-bound, bound, ...): low + (high - low) * u
                W += -bound
            else:
                np.multiply(u[:fi * fo], (2.0 / fi) ** 0.5, out=W)    # == rs.normal(0, sigma, ...): loc + scale * gauss
                W += 0.0
            W32 = by_shape[(fi, fo)] = W.reshape(fi, fo).astype(np.float32)
        else:
            W32 = W32.copy()
        b = np.zeros(fo) if bias_strategy == "zeros" else np.ones(fo)
        out.append((W32, b.astype(np.float32)))
    return out


class DenseLayerView(object):
    """One dense layer: ``fan_in``, ``fan_out``, activation name and live numpy views ``w_`` / ``b_``."""

    def __init__(self, owner, index, fan_in, fan_out, activation):
        self._owner, self._index = owner, index
        self.fan_in, self.fan_out, self.activation = fan_in, fan_out, activation

    @property
    def w_(self):
        return self._owner._layer_params(self._index)[0]

    @property
    def b_(self):
        return self._owner._layer_params(self._index)[1]

    def __repr__(self):
        return "DenseLayerView(fan_in=%d, fan_out=%d, activation=%r)" % (self.fan_in, self.fan_out, self.activation)


class VAETopography(object):
    """encoder -> (z_mean, z_log_sigma) heads -> reparameterisation -> decoder with a sigmoid output."""

    def __init__(self, n_hidden, input_shape, activation, n_latent_factors, layer_type="xavier",
                 bias_strategy="zeros", random_state=None, params_source=None):
        self.n_latent_factors = resolve_latent(n_latent_factors, input_shape)
        hidden = normalize_hidden(n_hidden)
        if layer_type not in PERMITTED_LAYER_TYPES:
            raise ValueError("layer_type must be one of %r" % list(PERMITTED_LAYER_TYPES))
        if bias_strategy not in PERMITTED_BIAS_STRATEGIES:
            raise ValueError("bias_strategy must be one of %r" % list(PERMITTED_BIAS_STRATEGIES))
        self.random_state = get_random_state(random_state)
        self.hidden, self.input_shape = hidden, int(input_shape)
        self.activation, self.layer_type, self.bias_strategy = activation, layer_type, bias_strategy
        widths = [self.input_shape] + hidden
        enc = list(zip(widths[:-1], widths[1:]))
        dec = [(self.n_latent_factors, widths[-1])] + [(o, i) for i, o in reversed(enc)]   # layer.py:293-297
        heads = [(widths[-1], self.n_latent_factors)] * 2
        self.dims = enc + heads + dec                      # ABI order
        self._params_source = params_source
        L = len(enc)
        self.encode_layers_ = [DenseLayerView(self, i, fi, fo, activation) for i, (fi, fo) in enumerate(enc)]
        self.z_mean_layer_ = DenseLayerView(self, L, heads[0][0], heads[0][1], activation)
        self.z_log_sigma_layer_ = DenseLayerView(self, L + 1, heads[1][0], heads[1][1], activation)
        self.decode_layers_ = [
            DenseLayerView(self, L + 2 + j, fi, fo, "sigmoid" if j == len(dec) - 1 else activation)   # layer.py:333-338
            for j, (fi, fo) in enumerate(dec)]
        # fan-ins of the encoder, fan-ins of the decoder, output width (layer.py:60-72)
        self.shape = tuple([fi for fi, _ in enc] + [fi for fi, _ in dec] + [dec[-1][1]])

    def initial_parameters(self):
        return initial_parameters(self.dims, self.layer_type, self.bias_strategy, self.random_state.seed_value)

    def split_flat(self, flat):
        out, o = [], 0
        for fi, fo in self.dims:
            W = np.asarray(flat[o:o + fi * fo]).reshape(fi, fo); o += fi * fo
            b = np.asarray(flat[o:o + fo]); o += fo
            out.append((W, b))
        if o != len(flat):
            raise ValueError("flat parameter vector has %d entries, expected %d" % (len(flat), o))
        return out

    @staticmethod
    def join_flat(params):
        return np.concatenate([np.concatenate([np.asarray(W, dtype=np.float32).ravel(),
                                               np.asarray(b, dtype=np.float32).ravel()]) for W, b in params])

    def _layer_params(self, index):
        if self._params_source is None:
            raise RuntimeError("topography is not attached to a fitted estimator")
        return self._params_source()[index]

    def get_weights_biases(self):
        """(weights, biases) of the encode + decode layers -- the latent heads are not included (layer.py:83-96)."""
        layers = self.encode_layers_ + self.decode_layers_
        return [tuple(l.w_ for l in layers), tuple(l.b_ for l in layers)]
