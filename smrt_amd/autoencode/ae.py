"""``AutoEncoder`` -- the plain symmetric auto-encoder on the HIP library.

Drop-in for the reference class (smrt/autoencode/autoencoder.py:91-282 on smrt/autoencode/base.py:44-258): same
constructor (18 arguments, no latent layer), ``fit`` / ``transform`` = ``encode`` / ``feed_forward`` (= reconstruct),
fitted attributes and error conventions.  It runs on the same kernels as the variational model: the C ABI handle is
created with ``SMRT_MODEL_AE`` (loss = mean over all elements of the squared error, the user's activation on every
layer including the output, every layer regularised; layer.py:163-190, autoencoder.py:260).
"""
from __future__ import annotations

import atexit
import weakref

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils.validation import check_array, check_is_fitted

from .. import _hip
from ..utils import DEFAULT_DROPOUT, DEFAULT_L2, NPDTYPE, get_random_state
from .topography import (PERMITTED_BIAS_STRATEGIES, PERMITTED_LAYER_TYPES, DenseLayerView, VAETopography,
                         initial_parameters, normalize_hidden)
from .vae import _UNCONSTRUCTIBLE, _bounded_float, _check_activation_optimizer, _close_handle, _positive_int

__all__ = ["AutoEncoder", "AETopography"]


class AETopography(object):
    """encoder pyramid and its mirror; ``shape`` = encoder fan-ins, decoder fan-ins, output width (layer.py:60-72)."""

    def __init__(self, n_hidden, input_shape, activation, layer_type="xavier", bias_strategy="zeros", random_state=None,
                 params_source=None):
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
        dec = [(o, i) for i, o in reversed(enc)]                   # layer.py:165
        self.dims = enc + dec
        self._params_source = params_source
        self.encode_layers_ = [DenseLayerView(self, i, fi, fo, activation) for i, (fi, fo) in enumerate(enc)]
        self.decode_layers_ = [DenseLayerView(self, len(enc) + j, fi, fo, activation) for j, (fi, fo) in enumerate(dec)]
        self.shape = tuple([fi for fi, _ in enc] + [fi for fi, _ in dec] + [dec[-1][1]])

    def initial_parameters(self):
        return initial_parameters(self.dims, self.layer_type, self.bias_strategy, self.random_state.seed_value)

    split_flat = VAETopography.split_flat
    join_flat = staticmethod(VAETopography.join_flat)
    _layer_params = VAETopography._layer_params
    get_weights_biases = VAETopography.get_weights_biases


class AutoEncoder(BaseEstimator, TransformerMixin):
    """Symmetric auto-encoder minimising the mean squared reconstruction error, trained on one MI355X.

    Parameters are those of the reference, same order and defaults (autoencoder.py:210-213).  ``fit`` keeps the
    reference's schedule (contiguous unshuffled mini-batches, ragged tail, ``epoch_costs_`` = cost of each epoch's
    last batch, optional early stopping; base.py:198-240).
    """

    def __init__(self, n_hidden, activation_function="sigmoid", learning_rate=0.05, n_epochs=20, batch_size=128,
                 min_change=1e-3, verbose=0, display_step=5, learning_function="rms_prop", early_stopping=False,
                 bias_strategy="zeros", random_state=None, layer_type="xavier", dropout=DEFAULT_DROPOUT,
                 l2_penalty=DEFAULT_L2, gclip_min=-5.0, gclip_max=5.0, clip=True):
        self.n_hidden = n_hidden
        self.activation_function = activation_function
        self.learning_rate = learning_rate
        self.n_epochs = n_epochs
        self.batch_size = batch_size
        self.min_change = min_change
        self.verbose = verbose
        self.display_step = display_step
        self.learning_function = learning_function
        self.early_stopping = early_stopping
        self.bias_strategy = bias_strategy
        self.random_state = random_state
        self.layer_type = layer_type
        self.dropout = dropout
        self.l2_penalty = l2_penalty
        self.gclip_min = gclip_min
        self.gclip_max = gclip_max
        self.clip = clip
        atexit.register(_close_handle, weakref.ref(self))       # base.py:75-77

    def _validate_for_fit(self):                                # base.py:79-90
        self.min_change = _bounded_float(self, "min_change")
        self.learning_rate = _bounded_float(self, "learning_rate")
        self.verbose = _positive_int(self, "verbose")
        self.display_step = _positive_int(self, "display_step")
        self.n_epochs = _positive_int(self, "n_epochs")
        self.dropout = _bounded_float(self, "dropout", ltet=True)
        if self.l2_penalty is not None:
            self.l2_penalty = _bounded_float(self, "l2_penalty")

    def clean_session(self):
        sess = self.__dict__.pop("sess", None)
        if sess is not None:
            sess.close()

    def _session(self):
        check_is_fitted(self, "topography_")
        sess = self.__dict__.get("sess")
        if sess is None or sess.closed:
            raise AttributeError("'AutoEncoder' object has no attribute 'sess' (session was cleaned)")
        return sess

    def set_initial_weights(self, params):
        self._initial_weights = None if params is None else [(np.array(W, np.float32), np.array(b, np.float32))
                                                             for W, b in params]
        return self

    def get_weights(self):
        return self.topography_.split_flat(self._session().get_params())

    def fit(self, X, y=None, **run_args):
        X = check_array(X, accept_sparse=False, ensure_2d=True, dtype=NPDTYPE)
        seeded_random_state = get_random_state(self.random_state)
        n_samples, n_features = X.shape
        self._validate_for_fit()
        activation, learning = _check_activation_optimizer(self.activation_function, self.learning_function)
        topo = AETopography(n_hidden=self.n_hidden, input_shape=n_features, activation=activation,
                            layer_type=self.layer_type, bias_strategy=self.bias_strategy,
                            random_state=seeded_random_state, params_source=self.get_weights)
        if learning in _UNCONSTRUCTIBLE:
            raise TypeError("__init__() missing 1 required positional argument: %r" % _UNCONSTRUCTIBLE[learning])
        device = getattr(self, "device", None)
        if device is None:
            import torch
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.clean_session()
        batch_size = int(self.batch_size)
        sess = _hip.VaeHandle(n_features=n_features, hidden=topo.hidden, n_latent=0, activation=activation,
                              optimizer=learning, learning_rate=self.learning_rate, l2_penalty=self.l2_penalty,
                              clip=self.clip, gclip_min=self.gclip_min, gclip_max=self.gclip_max, eps=1e-10,
                              dropout=self.dropout, seed=seeded_random_state.seed_value,
                              max_batch=max(1, min(batch_size, n_samples)), device=device, model="ae")
        self.sess = sess
        init = getattr(self, "_initial_weights", None)
        sess.set_params(AETopography.join_flat(init if init is not None else topo.initial_parameters()))
        self.topography_ = topo
        if self.n_epochs == 0:
            raise UnboundLocalError("local variable 'c' referenced before assignment")      # base.py:243
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        X_dev = _hip.to_device_padded(X, sess.device)
        costs, ms = sess.train_epochs(X_dev, batch_size, self.n_epochs, early_stopping=bool(self.early_stopping),
                                      min_change=self.min_change)
        epoch_times = [float(t) / 1e3 for t in ms]
        for epoch, (c, t) in enumerate(zip(costs, epoch_times)):
            if epoch % self.display_step == 0 and self.verbose > 1:
                print("Epoch: %i, cost=%.6f, time=%.4f (sec)" % (epoch + 1, c, t))
        self.train_cost_ = costs[-1]
        self.epoch_times_ = epoch_times
        self.epoch_costs_ = list(costs)
        self.random_state_ = seeded_random_state
        if self.verbose:
            print("Optimization complete after %i epoch(s). Average epoch time: %.4f seconds"
                  % (len(epoch_times), np.average(epoch_times)))
        return self

    def _check_X(self, X):
        sess = self._session()
        X = np.asarray(X, dtype=np.float32)
        if X.ndim != 2 or X.shape[1] != sess.n_features:
            raise ValueError("Cannot feed value of shape %r for Tensor of shape (?, %d)" % (X.shape, sess.n_features))
        return sess, X

    def transform(self, X):
        """``encode(X)`` (base.py:103-113)."""
        return self.encode(X)

    def encode(self, X):
        """The compressed representation, fp32 ``[n, n_hidden[-1]]`` (autoencoder.py:262-266)."""
        sess, X = self._check_X(X)
        return sess.encode(X)

    def feed_forward(self, X):
        """Encode then decode: the reconstruction of ``X`` (autoencoder.py:268-282)."""
        sess, X = self._check_X(X)
        return sess.decode(sess.encode(X, to_host=False))
