"""``VariationalAutoEncoder`` -- sklearn estimator whose arithmetic runs in the HIP library.

Drop-in for the reference class (smrt/autoencode/autoencoder.py:285-572 on top of
smrt/autoencode/base.py:44-258): same constructor, methods, fitted attributes and error
conventions.  The four ``Session.run`` seams map onto ``include/smrt_hip.h`` entry points:

    base.py:190      init             -> smrt_vae_create + smrt_vae_set_params
    base.py:215-217  train step loop  -> smrt_vae_train_epochs  (whole _train loop on device)
    autoencoder.py:500  encode        -> smrt_vae_encode
    autoencoder.py:517  decode        -> smrt_vae_decode
"""
from __future__ import annotations

import atexit
import weakref

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils.validation import check_array, check_is_fitted

from .. import _hip
from ..utils import DEFAULT_DROPOUT, DEFAULT_L2, NPDTYPE, get_random_state, validate_float
from .data_parallel import DataParallelTrainer, dist_info, local_row_index
from .topography import VAETopography

def fit_group(estimators, Xs, n_threads=8):
    """``est.fit(X)`` for every pair, with the training loops run SIDE BY SIDE where the library can group them.

    The per-class VAEs of one ``smrt_balance`` call are independent jobs of the same shape whose default-batch steps keep
    4 to 120 of the GPU's 256 CUs busy, ten dependent launches a step.  ``smrt_vae_train_epochs_multi`` issues the launches
    of one kernel for all of them as one grouped launch -- the same kernels on the same numbers (results are bit-identical
    to separate fits), an eighth of the launches.  Estimators that do not qualify (early stopping, data-parallel, different
    schedules, nets on the LDS-resident paths) are fitted one by one.  ``epoch_times_`` of grouped fits are the shared
    epochs' times."""
    estimators, Xs = list(estimators), list(Xs)
    if not estimators:
        return []
    import os
    first = estimators[0]
    groupable = os.environ.get("SMRT_CLASS_GROUP", "1") != "0" and len(estimators) > 1 and all(
        not e.early_stopping and not getattr(e, "data_parallel", False) and int(e.batch_size) == int(first.batch_size)
        and int(e.n_epochs) == int(first.n_epochs) and getattr(e, "device", None) == getattr(first, "device", None)
        for e in estimators)
    if not groupable:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(int(n_threads), len(estimators)))) as pool:
            return list(pool.map(lambda ex: ex[0].fit(ex[1]), zip(estimators, Xs)))
    import sys
    import time
    timing = os.environ.get("SMRT_TIMING") == "1"
    t0 = time.perf_counter()
    # sessions, initial weights and uploads of the classes side by side (numpy and ctypes release the GIL for most of it)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(int(n_threads), len(estimators)))) as pool:
        prepared = list(pool.map(lambda ex: ex[0]._prepare_fit(ex[1]), zip(estimators, Xs)))
        t_mid = time.perf_counter()
        X_devs = list(pool.map(lambda ex: _hip.to_device_padded(ex[1][0], ex[0].sess.device), zip(estimators, prepared)))
    t1 = time.perf_counter()
    if timing:
        sys.stderr.write("[fit_group] sessions + initial weights %.1f ms, uploads %.1f ms\n" % (1e3 * (t_mid - t0), 1e3 * (t1 - t_mid)))
    res = _hip.train_epochs_multi([e.sess for e in estimators], X_devs, prepared[0][2], int(first.n_epochs))
    if timing:
        sys.stderr.write("[fit_group] prepare %.1f ms, train %.1f ms (%s)\n" % (
            1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1),
            "declined" if res is None else "grouped, %d epochs of %.3f ms on the device" % (len(res[1]), float(np.mean(res[1])))))
    if res is None:
        # declined (LDS-resident nets, big batches): every handle trains alone, side by side on its own stream and host
        # thread (ctypes releases the GIL).  The persistent whole-epoch launch and the two-launch step agree to 2e-5 only,
        # and a fixed random_state must give a fixed result whatever the thread timing, the number of classes or the
        # number of ranks: so EVERY class whose net qualifies takes the persistent launch, and the device's budget of
        # co-resident workgroups only decides how many of them run at the same time (they start in class order; a class
        # waits for room instead of changing its path).  Only a class that alone exceeds the budget keeps the other step.
        import threading
        budget = _hip.persistent_budget(estimators[0].sess.device.index or 0)
        needs = []
        for e, p in zip(estimators, prepared):
            need = e.sess.persistent_workgroups(min(int(p[2]), int(p[0].shape[0])))
            grant = 0 < need <= budget
            e.sess.set_persistent(1 if grant else 0)
            needs.append(need if grant else 0)
        room = threading.Condition()
        state = {"free": budget, "turn": 0}

        def train(k):
            est, X_dev, prep = estimators[k], X_devs[k], prepared[k]
            with room:                           # start in class order, when the workgroups fit beside the running ones
                room.wait_for(lambda: state["turn"] == k and state["free"] >= needs[k])
                state["free"] -= needs[k]
                state["turn"] += 1
                room.notify_all()
            try:
                return est.sess.train_epochs(X_dev, prep[2], est.n_epochs)
            finally:
                with room:
                    state["free"] += needs[k]
                    room.notify_all()

        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(estimators)) as pool:
            each = list(pool.map(train, range(len(estimators))))
    else:
        each = [(res[0][k], res[1]) for k in range(len(estimators))]
    for e, p, (costs, ms) in zip(estimators, prepared, each):
        e._finish_fit(costs, [float(t) / 1e3 for t in ms], p[1])
    return estimators


__all__ = ["VariationalAutoEncoder", "PERMITTED_ACTIVATIONS", "PERMITTED_OPTIMIZERS"]

PERMITTED_ACTIVATIONS = ("elu", "identity", "relu", "sigmoid", "tanh")            # autoencoder.py:28-34
PERMITTED_OPTIMIZERS = ("adadelta", "adagrad", "adagrad-da", "adam", "momentum", "proximal-sgd",
                        "proximal-adagrad", "rms_prop", "sgd")                       # autoencoder.py:38-48
# tf.train.MomentumOptimizer / AdagradDAOptimizer need a second positional argument, so the reference's
# learning_function(rate) call (base.py:138,141) raises TypeError for these two keys.
_UNCONSTRUCTIBLE = {"momentum": "momentum", "adagrad-da": "global_step"}


def _positive_int(instance, name):
    """int(value) >= 0, bools rejected (base.py:24-32)."""
    val = getattr(instance, name)
    try:
        if isinstance(val, (bool, np.bool_)):
            raise ValueError
        res = int(val)
        if res < 0:
            raise ValueError
    except Exception:
        raise ValueError("%s must be an int >= 0" % name)
    return res


def _bounded_float(instance, name, upper_bound=1.0, ltet=False):
    """float(value) in (0, upper) -- or (0, upper] with ``ltet`` (base.py:35-41)."""
    res = float(getattr(instance, name))
    validate_float(res, name, upper_bound, ltet=ltet)
    return res


def _check_activation_optimizer(activation_function, learning_function):
    """String keys only; unknown key -> ValueError, non-string -> TypeError (autoencoder.py:51-88)."""
    if not isinstance(activation_function, str):
        raise TypeError("Activation function must be a string")
    if activation_function not in PERMITTED_ACTIVATIONS:
        raise ValueError("Permitted activation functions: %r" % list(PERMITTED_ACTIVATIONS))
    if not isinstance(learning_function, str):
        raise TypeError("Learning function must be a string")
    if learning_function not in PERMITTED_OPTIMIZERS:
        raise ValueError("Permitted learning functions: %r" % list(PERMITTED_OPTIMIZERS))
    return activation_function, learning_function


def _close_handle(ref):
    est = ref()
    if est is not None:
        est.clean_session()


class VariationalAutoEncoder(BaseEstimator, TransformerMixin):
    """Variational auto-encoder trained on one MI355X.

    Loss: ``mean_b(BCE_b + KL_b) + l2 * sum_{enc,dec layers} 1/2 |W|^2`` with a sigmoid output layer
    (autoencoder.py:469-480).  ``fit`` runs the reference's training schedule -- contiguous unshuffled
    mini-batches with a ragged tail, ``epoch_costs_`` holding the cost of each epoch's LAST batch,
    optional early stopping on ``|last - c| <= min_change`` (base.py:198-240) -- entirely on the device.

    Parameters are those of the reference, in the same order with the same defaults
    (autoencoder.py:416-420); ``n_hidden`` (int or list) and ``n_latent_factors`` (int, or a float in
    (0, 1) meaning a fraction of the input width) are required.

    Additional, optional hooks (not in the reference): ``set_initial_weights`` /
    ``get_weights`` / ``set_weights`` for checkpointing and parity injection, the ``device``
    attribute (GPU ordinal, default: current torch device), and the ``data_parallel`` attribute: when it is
    true and ``torch.distributed`` is initialised with more than one rank, ``fit`` (called by EVERY rank with
    the same ``X``) trains one replicated model on row stripes of each mini-batch with a gradient all-reduce
    per step (``smrt_amd.autoencode.data_parallel``); the fitted estimators are identical on all ranks.
    """

    def __init__(self, n_hidden, n_latent_factors, activation_function="sigmoid", learning_rate=0.05,
                 n_epochs=20, batch_size=128, min_change=1e-3, verbose=0, display_step=5,
                 learning_function="rms_prop", early_stopping=False, bias_strategy="zeros", random_state=None,
                 layer_type="xavier", dropout=DEFAULT_DROPOUT, l2_penalty=DEFAULT_L2, eps=1e-10,
                 gclip_min=-5.0, gclip_max=5.0, clip=True):
        self.n_hidden = n_hidden
        self.n_latent_factors = n_latent_factors
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
        self.eps = eps
        self.gclip_min = gclip_min
        self.gclip_max = gclip_max
        self.clip = clip
        atexit.register(_close_handle, weakref.ref(self))       # base.py:75-77

    # ------------------------------------------------------------------ validation (base.py:79-90)
    def _validate_for_fit(self):
        self.min_change = _bounded_float(self, "min_change")
        self.learning_rate = _bounded_float(self, "learning_rate")
        self.verbose = _positive_int(self, "verbose")
        self.display_step = _positive_int(self, "display_step")
        self.n_epochs = _positive_int(self, "n_epochs")
        self.dropout = _bounded_float(self, "dropout", ltet=True)
        if self.l2_penalty is not None:
            self.l2_penalty = _bounded_float(self, "l2_penalty")

    # ------------------------------------------------------------------ session lifecycle
    def clean_session(self):
        """Release the device handle (reference: close the TF session, base.py:115-121)."""
        sess = self.__dict__.pop("sess", None)
        if sess is not None:
            sess.close()

    def _session(self):
        check_is_fitted(self, "topography_")
        sess = self.__dict__.get("sess")
        if sess is None or sess.closed:
            raise AttributeError("'VariationalAutoEncoder' object has no attribute 'sess' (session was cleaned)")
        return sess

    # ------------------------------------------------------------------ checkpoint / injection hooks
    def set_initial_weights(self, params):
        """Use ``params`` ([(W, b), ...] in layer order enc.., z_mean, z_log_sigma, dec..) at the next ``fit``."""
        self._initial_weights = None if params is None else [(np.array(W, np.float32), np.array(b, np.float32))
                                                             for W, b in params]
        return self

    def get_weights(self):
        return self.topography_.split_flat(self._session().get_params())

    def set_weights(self, params):
        self._session().set_params(VAETopography.join_flat(params))
        return self

    # ------------------------------------------------------------------ fit (base.py:151-254, autoencoder.py:439-486)
    def fit(self, X, y=None, **run_args):
        """Train on ``X`` (values in [0, 1]: the loss is Bernoulli).  ``run_args`` is accepted for
        signature compatibility (the reference forwards it to ``Session.run``) and ignored."""
        X, seeded_random_state, batch_size = self._prepare_fit(X)
        sess = self.sess
        n_samples = X.shape[0]
        rank, world, dist = dist_info()
        if getattr(self, "data_parallel", False) and world > 1:
            # one replicated model, each rank keeps only its stripes of every mini-batch in HBM
            idx = local_row_index(n_samples, batch_size, world, rank)
            X_dev = _hip.to_device_padded(X[idx], sess.device)
            trainer = DataParallelTrainer(sess, X_dev, n_samples, batch_size, rank, world, dist)
            costs, epoch_times = trainer.run_epochs(self.n_epochs, early_stopping=bool(self.early_stopping),
                                                    min_change=self.min_change)
        else:
            X_dev = _hip.to_device_padded(X, sess.device)         # fp64 -> fp32 feed cast, once (base.py:174)
            costs, ms = sess.train_epochs(X_dev, batch_size, self.n_epochs, early_stopping=bool(self.early_stopping),
                                          min_change=self.min_change)
            epoch_times = [float(t) / 1e3 for t in ms]
        return self._finish_fit(costs, epoch_times, seeded_random_state)

    def _prepare_fit(self, X):
        """Everything of ``fit`` in front of the training loop: validation, topography, session, initial weights."""
        # fp32 rows stay fp32: the reference converts to fp64 here (base.py:174) and feeds the graph fp32 -- a lossless round
        # trip, but 8 x 31 MB of copies under the GIL when smrt_balance prepares C3's classes side by side
        X = check_array(X, accept_sparse=False, ensure_2d=True, dtype=[NPDTYPE, np.float32])
        seeded_random_state = get_random_state(self.random_state)
        n_samples, n_features = X.shape

        self._validate_for_fit()
        eps = _bounded_float(self, "eps")
        activation, learning = _check_activation_optimizer(self.activation_function, self.learning_function)

        topo = VAETopography(n_hidden=self.n_hidden, input_shape=n_features, activation=activation,
                             n_latent_factors=self.n_latent_factors, layer_type=self.layer_type,
                             bias_strategy=self.bias_strategy, random_state=seeded_random_state,
                             params_source=self.get_weights)
        if learning in _UNCONSTRUCTIBLE:
            raise TypeError("__init__() missing 1 required positional argument: %r" % _UNCONSTRUCTIBLE[learning])

        device = getattr(self, "device", None)
        if device is None:
            import torch
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.clean_session()                                      # refit replaces the session (base.py:186)
        batch_size = int(self.batch_size)
        sess = _hip.VaeHandle(n_features=n_features, hidden=topo.hidden, n_latent=topo.n_latent_factors,
                              activation=activation, optimizer=learning, learning_rate=self.learning_rate,
                              l2_penalty=self.l2_penalty, clip=self.clip, gclip_min=self.gclip_min,
                              gclip_max=self.gclip_max, eps=eps, dropout=self.dropout,
                              seed=seeded_random_state.seed_value, max_batch=max(1, min(batch_size, n_samples)),
                              device=device)
        self.sess = sess
        init = getattr(self, "_initial_weights", None)
        sess.set_params(VAETopography.join_flat(init if init is not None else topo.initial_parameters()))
        self.topography_ = topo

        if self.n_epochs == 0:
            # the reference reads the loop variable `c` after a loop that never ran (base.py:243)
            raise UnboundLocalError("local variable 'c' referenced before assignment")
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1")
        return X, seeded_random_state, batch_size

    def _finish_fit(self, costs, epoch_times, seeded_random_state):
        """Everything of ``fit`` behind the training loop: progress lines and the fitted attributes (base.py:220-252)."""
        for epoch, (c, t) in enumerate(zip(costs, epoch_times)):
            if epoch % self.display_step == 0 and self.verbose > 1:
                print("Epoch: %i, cost=%.6f, time=%.4f (sec)" % (epoch + 1, c, t))
        if self.early_stopping and self.verbose and len(costs) < self.n_epochs:
            print("Convergence reached at epoch %i, stopping early" % (len(costs) - 1))

        self.train_cost_ = costs[-1]
        self.epoch_times_ = epoch_times
        self.epoch_costs_ = list(costs)
        self.random_state_ = seeded_random_state                  # used later by generate (base.py:248)
        if self.verbose:
            print("Optimization complete after %i epoch(s). Average epoch time: %.4f seconds"
                  % (len(epoch_times), np.average(epoch_times)))
        return self

    # ------------------------------------------------------------------ inference
    def transform(self, X):
        """``encode(X)`` (base.py:103-113)."""
        return self.encode(X)

    def encode(self, X):
        """Inferential network: returns ``[z_mean, z_log_sigma]``, two fp32 arrays (autoencoder.py:488-500)."""
        sess = self._session()
        X = np.asarray(X, dtype=np.float32)
        if X.ndim != 2 or X.shape[1] != sess.n_features:
            raise ValueError("Cannot feed value of shape %r for Tensor of shape (?, %d)" % (X.shape, sess.n_features))
        mu, ls = sess.encode(X)
        return [mu, ls]

    def feed_forward(self, X):
        """Generative network applied to latent rows ``X`` (autoencoder.py:502-517)."""
        sess = self._session()
        X = check_array(X, dtype=NPDTYPE, ensure_2d=False)
        if X.ndim != 2 or X.shape[1] != sess.n_latent:
            raise ValueError("Cannot feed value of shape %r for Tensor of shape (?, %d)" % (X.shape, sess.n_latent))
        return sess.decode(X.astype(np.float32))

    def generate_from_sample(self, X, **nrm_args):
        """Encode ``X``, perturb with ``eps ~ N(loc, scale)`` drawn from ``random_state_``, decode
        (autoencoder.py:519-547)."""
        check_is_fitted(self, "topography_")
        X = check_array(X, dtype=NPDTYPE, ensure_2d=True, accept_sparse=False)
        z_mu, log_sigma = self.encode(X)
        eps = _hip.rng_normal(self.random_state_.state, log_sigma.shape, **nrm_args)         # numpy's legacy stream, bit for bit
        z_mu += eps * np.exp(log_sigma)           # fp64 product stored into the fp32 array, as the reference
        return self.feed_forward(z_mu)

    def generate(self, n=1, **nrm_args):
        """Decode ``n`` draws ``z ~ N(loc, scale)`` from ``random_state_`` (autoencoder.py:549-572)."""
        check_is_fitted(self, "topography_")
        z = _hip.rng_normal(self.random_state_.state, (n, self.n_latent_factors), **nrm_args)
        return self.feed_forward(z)
