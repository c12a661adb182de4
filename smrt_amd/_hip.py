"""ctypes binding of ``include/smrt_hip.h`` (``smrt_amd/_lib/libsmrt_hip.so``).

PyTorch-ROCm is used only to own device memory and hand ``data_ptr()`` across; every
number is produced by the HIP library.  There is NO CPU fallback: a missing library or
a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB_PATH = os.environ.get("SMRT_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libsmrt_hip.so")
_lib = None

MAX_HIDDEN = 6

ACTIVATION_IDS = {"elu": 0, "identity": 1, "relu": 2, "sigmoid": 3, "tanh": 4}
OPTIMIZER_IDS = {"sgd": 0, "rms_prop": 1, "adam": 2, "adagrad": 3, "adadelta": 4,
                 "proximal-sgd": 5, "proximal-adagrad": 6}

EXPORTS = [
    "smrt_vae_create", "smrt_vae_destroy", "smrt_vae_param_count", "smrt_vae_set_params",
    "smrt_vae_get_params", "smrt_vae_get_opt_state", "smrt_vae_reset_opt_state", "smrt_vae_step_count",
    "smrt_vae_set_dropout_masks", "smrt_vae_train_step", "smrt_vae_train_epochs", "smrt_vae_train_epochs_multi",
    "smrt_vae_dp_count", "smrt_persistent_budget", "smrt_vae_persistent_workgroups", "smrt_vae_set_persistent", "smrt_vae_persistent_form",
    "smrt_vae_grad_step", "smrt_vae_apply_grads", "smrt_vae_grad_step_async", "smrt_vae_apply_grads_async",
    "smrt_vae_set_stream", "smrt_vae_encode", "smrt_vae_decode",
    "smrt_vae_generate", "smrt_vae_generate_noalloc", "smrt_vae_generate_noalloc_max_rows", "smrt_vae_sample_eps", "smrt_vae_last_epoch_launches", "smrt_vae_profile_epochs",
    "smrt_assemble_rows", "smrt_assemble_rows_part", "smrt_assemble_rows_synth_range", "smrt_assemble_rows_f64", "smrt_all_finite", "smrt_rng_normal", "smrt_rng_permutation", "smrt_knn_f64", "smrt_smote_rows_f64", "smrt_diag_mfma_peak", "smrt_diag_mfma_peak_16x16x4", "smrt_diag_hbm_bandwidth", "smrt_diag_step_floor",
    "smrt_last_error", "smrt_version",
]


class VaeCfg(C.Structure):
    _fields_ = [
        ("n_features", C.c_int32), ("n_hidden_layers", C.c_int32), ("hidden", C.c_int32 * MAX_HIDDEN),
        ("n_latent", C.c_int32), ("activation", C.c_int32), ("optimizer", C.c_int32),
        ("learning_rate", C.c_float), ("l2_penalty", C.c_float), ("clip", C.c_int32),
        ("gclip_min", C.c_float), ("gclip_max", C.c_float), ("bce_lo", C.c_float), ("bce_hi", C.c_float),
        ("dropout_keep", C.c_float), ("seed", C.c_uint64), ("max_batch", C.c_int32), ("model", C.c_int32),
    ]


class HipLibraryError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryError(
            "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C smrt_amd/csrc`. There is no CPU fallback." % _LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be
    # in the process BEFORE this library is mapped so that both resolve to the same runtime (one HSA
    # client, shared device pointers); loading /opt/rocm's copy next to it leaves the second one blind.
    import torch  # noqa: F401
    lib = C.CDLL(_LIB_PATH)
    vp, f32p, i64, i32 = C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int32
    lib.smrt_last_error.restype = C.c_char_p
    lib.smrt_version.restype = C.c_char_p
    lib.smrt_vae_create.argtypes = [C.POINTER(VaeCfg), C.c_int, C.POINTER(vp)]
    lib.smrt_vae_destroy.argtypes = [vp]
    dp = C.POINTER(C.c_double)
    lib.smrt_knn_f64.argtypes = [vp, i64, i64, vp, i64, i64, i64, i32, vp, vp, vp]
    lib.smrt_smote_rows_f64.argtypes = [vp, i64, i64, i64, vp, vp, vp, i64, i32, vp, i64, vp]
    lib.smrt_diag_mfma_peak.argtypes = [i32, dp]
    lib.smrt_diag_mfma_peak_16x16x4.argtypes = [i32, dp]
    lib.smrt_diag_hbm_bandwidth.argtypes = [i32, i64, dp, dp]
    lib.smrt_diag_step_floor.argtypes = [i32, i32, i32, i32, i32, dp]
    lib.smrt_vae_destroy.restype = None
    lib.smrt_vae_param_count.argtypes = [vp]; lib.smrt_vae_param_count.restype = i64
    lib.smrt_vae_dp_count.argtypes = [vp]; lib.smrt_vae_dp_count.restype = i64
    lib.smrt_vae_step_count.argtypes = [vp]; lib.smrt_vae_step_count.restype = i64
    lib.smrt_vae_last_epoch_launches.argtypes = [vp]
    lib.smrt_vae_set_params.argtypes = [vp, f32p, i64]
    lib.smrt_vae_get_params.argtypes = [vp, f32p, i64]
    lib.smrt_vae_get_opt_state.argtypes = [vp, C.c_int, f32p, i64]
    lib.smrt_vae_reset_opt_state.argtypes = [vp]
    lib.smrt_vae_set_dropout_masks.argtypes = [vp, f32p, i64]
    lib.smrt_vae_train_step.argtypes = [vp, vp, i64, i32, vp, f32p, f32p]
    lib.smrt_vae_train_epochs.argtypes = [vp, vp, i64, i64, i32, i32, i32, C.c_float, f32p, f32p, C.POINTER(i32)]
    lib.smrt_vae_train_epochs_multi.argtypes = [C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), i32, i32,
                                                f32p, f32p]
    lib.smrt_persistent_budget.argtypes = [i32]; lib.smrt_persistent_budget.restype = i32
    lib.smrt_vae_persistent_workgroups.argtypes = [vp, i32]; lib.smrt_vae_persistent_workgroups.restype = i32
    lib.smrt_vae_persistent_form.argtypes = [vp]; lib.smrt_vae_persistent_form.restype = i32
    lib.smrt_vae_set_persistent.argtypes = [vp, i32]
    lib.smrt_vae_grad_step.argtypes = [vp, vp, i64, i32, i64, i64, vp, vp]
    lib.smrt_vae_apply_grads.argtypes = [vp, vp, f32p]
    lib.smrt_vae_grad_step_async.argtypes = [vp, vp, i64, i32, i64, i64, vp, vp]
    lib.smrt_vae_apply_grads_async.argtypes = [vp, vp, vp]
    lib.smrt_vae_set_stream.argtypes = [vp, vp, i32]
    lib.smrt_vae_encode.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.smrt_vae_decode.argtypes = [vp, vp, i64, i64, vp, i64]
    lib.smrt_vae_generate.argtypes = [vp, i64, C.c_uint64, C.c_float, C.c_float, vp, i64]
    lib.smrt_vae_generate_noalloc.argtypes = [vp, i64, C.c_uint64, C.c_float, C.c_float, vp, i64]
    lib.smrt_vae_generate_noalloc_max_rows.argtypes = [vp]; lib.smrt_vae_generate_noalloc_max_rows.restype = i64
    lib.smrt_vae_sample_eps.argtypes = [vp, i64, i64, i32, vp]
    lib.smrt_vae_profile_epochs.argtypes = [vp, vp, i64, i64, i32, i32, C.POINTER(C.c_double), C.POINTER(i64)]
    lib.smrt_assemble_rows.argtypes = [vp, i32, i64, i64, i64, vp, i64, i64, vp, i64, vp, i32]
    lib.smrt_assemble_rows_f64.argtypes = [vp, i32, i64, i64, i64, vp, i64, i64, vp, i64, vp, i32]
    lib.smrt_all_finite.argtypes = [vp, i32, i64, i32]
    lib.smrt_assemble_rows_part.argtypes = [vp, i32, i64, i64, i64, vp, i64, i64, vp, i64, vp, i32, i32]
    lib.smrt_assemble_rows_synth_range.argtypes = [vp, i64, i64, i64, i64, vp, i64, vp, i32, i64, i64]
    lib.smrt_rng_normal.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_double), C.c_double, C.c_double, vp, i64, i32]
    lib.smrt_rng_permutation.argtypes = [vp, C.POINTER(i32), vp, i64]
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise HipLibraryError("%s failed (%d): %s" % (what, rc, load().smrt_last_error().decode()))


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise HipLibraryError("no MI355X visible to PyTorch-ROCm; the HIP path has no CPU fallback")
    return torch


def rup4(v):
    return (int(v) + 3) & ~3


def sync_torch_stream(device):
    """Wait for the work PyTorch has queued on its current stream of ``device`` (uploads, fills) before the library's
    own stream touches the same memory.  Deliberately NOT ``torch.cuda.synchronize``: a device-wide synchronise is
    refused by the runtime while any stream of the process is being captured, and ``smrt_balance`` fits classes
    concurrently, each thread capturing its own epoch graph."""
    _torch().cuda.current_stream(device).synchronize()


def row_stride(d):
    """Leading dimension (floats) the wrapper gives device matrices of ``d`` columns: rows start on 128-byte boundaries
    once they are longer than that (a wave's 128-byte row pieces then touch one cache line each, not two), 16-byte
    boundaries otherwise.  The C ABI itself only asks for a multiple of 4."""
    d = int(d)
    return (d + 31) // 32 * 32 if d > 32 else rup4(d)


def to_device_padded(array, device):
    """Host [n, d] array -> contiguous fp32 device tensor [n, row_stride(d)]; returns (tensor, ld)."""
    torch = _torch()
    a = np.ascontiguousarray(array, dtype=np.float32)
    n, d = a.shape
    ld = row_stride(d)
    t = torch.zeros((n, ld), dtype=torch.float32, device=device)
    if n:
        t[:, :d] = torch.from_numpy(a).to(device)
    sync_torch_stream(device)
    return t, ld


def diag_mfma_peak(device=0):
    """fp32 MFMA TFLOP/s a register-resident loop reaches on ``device`` (include/smrt_hip.h: diagnostics)."""
    _torch()
    out = C.c_double()
    _check(load().smrt_diag_mfma_peak(int(device), C.byref(out)), "smrt_diag_mfma_peak")
    return out.value


def diag_mfma_peak_16x16x4(device=0):
    """The same register-resident loop on v_mfma_f32_16x16x4_f32, the shape of the LDS-resident small-net kernels."""
    _torch()
    out = C.c_double()
    _check(load().smrt_diag_mfma_peak_16x16x4(int(device), C.byref(out)), "smrt_diag_mfma_peak_16x16x4")
    return out.value


def diag_hbm_bandwidth(device=0, nbytes=1 << 30):
    """(copy GB/s counting read + write, read-only GB/s) over ``nbytes`` of device memory."""
    _torch()
    c, r = C.c_double(), C.c_double()
    _check(load().smrt_diag_hbm_bandwidth(int(device), C.c_int64(int(nbytes)), C.byref(c), C.byref(r)),
           "smrt_diag_hbm_bandwidth")
    return c.value, r.value


def diag_step_floor(device=0, kernels_per_step=2, steps=154, workgroups=8, threads=512):
    """Microseconds per step of a captured chain of dependent one-word kernels: the floor under a mini-batch step."""
    _torch()
    out = C.c_double()
    _check(load().smrt_diag_step_floor(int(device), int(kernels_per_step), int(steps), int(workgroups), int(threads),
                                       C.byref(out)), "smrt_diag_step_floor")
    return out.value


def assemble_rows(X, synth, order, n_threads=None, synth_dtype=np.float32):
    """fp64 ``[X ; synth][order]`` in one multi-threaded pass (``order`` None: plain concatenation).

    ``X``: C-contiguous fp32/fp64 ``[n, d]``; ``synth``: ``[m, d]`` rows of ``synth_dtype`` (fp32: what the VAE kernels
    produce; fp64: SMOTE rows), may be ``None`` or have 0 rows (then this is a pure row gather); ``order``: int64 row
    numbers.  Host-only: needs the library, not a GPU."""
    lib = load()
    X = np.ascontiguousarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    n, d = X.shape
    synth_dtype = np.dtype(synth_dtype)
    if synth_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("synthetic rows must be fp32 or fp64")
    synth = np.empty((0, d), dtype=synth_dtype) if synth is None else np.ascontiguousarray(synth, dtype=synth_dtype)
    m = synth.shape[0]
    if m and synth.shape[1] != d:
        raise ValueError("synthetic rows have %d columns, X has %d" % (synth.shape[1], d))
    if order is not None:
        order = np.ascontiguousarray(order, dtype=np.int64)
    n_out = n + m if order is None else order.shape[0]
    out = np.empty((n_out, d), dtype=np.float64)
    if n_threads is None:
        n_threads = max(1, min(16, os.cpu_count() or 1))
    fn = lib.smrt_assemble_rows if synth_dtype == np.float32 else lib.smrt_assemble_rows_f64
    rc = fn(X.ctypes.data, int(X.dtype == np.float32), n, d, d, synth.ctypes.data if m else None, m, d,
            order.ctypes.data if order is not None else None, n_out, out.ctypes.data, int(n_threads))
    if rc != 0:
        raise HipLibraryError("smrt_assemble_rows rejected its arguments (%d)" % rc)
    return out


def all_finite(X, n_threads=None):
    """True when the C-contiguous fp32 / fp64 host array ``X`` holds no NaN or infinity (multi-threaded sweep)."""
    if n_threads is None:
        n_threads = max(1, min(16, os.cpu_count() or 1))
    rc = load().smrt_all_finite(X.ctypes.data, int(X.dtype == np.float32), X.size, int(n_threads))
    if rc < 0:
        raise HipLibraryError("smrt_all_finite rejected its arguments")
    return rc == 1


# ---- draws on numpy's legacy stream (host_rng.cpp): same numbers, same state afterwards, a fraction of the time
_RNG_MIN = 4096          # below this many values numpy's own loop is as fast as the round trip through get/set_state


def _legacy_state(rs):
    kind, key, pos, has_gauss, cached = rs.get_state(legacy=True)
    if kind != "MT19937":
        raise HipLibraryError("not a Mersenne-Twister RandomState")
    return np.ascontiguousarray(key, dtype=np.uint32).copy(), C.c_int32(int(pos)), C.c_int32(int(has_gauss)), C.c_double(float(cached))


def rng_normal(rs, size, loc=0.0, scale=1.0, n_threads=None):
    """``rs.normal(loc, scale, size)`` for a ``numpy.random.RandomState`` -- the same fp64 values bit for bit and the same
    generator state afterwards (the Mersenne-Twister pass stays sequential, the polar method's log / sqrt run on a thread
    pool).  Scalar ``loc`` / ``scale`` only; anything else, and small draws, go to numpy itself."""
    shape = (size,) if isinstance(size, (int, np.integer)) else tuple(size)
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    if (n < _RNG_MIN or not isinstance(rs, np.random.RandomState) or not np.isscalar(loc) or not np.isscalar(scale)
            or not np.isfinite(scale) or scale < 0):
        return rs.normal(loc, scale, size)
    key, pos, has, cached = _legacy_state(rs)
    out = np.empty(n, dtype=np.float64)
    if n_threads is None:
        n_threads = max(1, min(16, os.cpu_count() or 1))
    rc = load().smrt_rng_normal(key.ctypes.data, C.byref(pos), C.byref(has), C.byref(cached), float(loc), float(scale),
                                out.ctypes.data, n, int(n_threads))
    if rc != 0:
        raise HipLibraryError("smrt_rng_normal failed (%d: -1 bad argument, -2 out of host memory)" % rc)
    rs.set_state(("MT19937", key, int(pos.value), int(has.value), float(cached.value)))
    return out.reshape(shape)


def rng_permutation(rs, n):
    """``rs.permutation(np.arange(n))`` (int64) for a ``numpy.random.RandomState``, bit for bit, same state afterwards."""
    n = int(n)
    if n < _RNG_MIN or not isinstance(rs, np.random.RandomState):
        return rs.permutation(np.arange(n))
    key, pos, has, cached = _legacy_state(rs)
    out = np.empty(n, dtype=np.int64)
    rc = load().smrt_rng_permutation(key.ctypes.data, C.byref(pos), out.ctypes.data, n)
    if rc != 0:
        raise HipLibraryError("smrt_rng_permutation failed (%d: -1 bad argument, -2 out of host memory)" % rc)
    rs.set_state(("MT19937", key, int(pos.value), int(has.value), float(cached.value)))     # (a shuffle leaves the kept normal alone)
    return out


class SplitAssembly(object):
    """``assemble_rows`` in two halves over one preallocated fp64 output: ``originals()`` writes the rows whose source is
    ``X`` (callable while the synthetic rows are still being made), ``synthetic(synth)`` the rest; ``out`` is complete
    after both.  ``n_synth`` fp32 synthetic rows are expected; ``order`` as in ``assemble_rows``."""

    def __init__(self, X, n_synth, order, n_threads=None):
        self._lib = load()
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        self.X, self.n_synth = X, int(n_synth)
        self.order = None if order is None else np.ascontiguousarray(order, dtype=np.int64)
        n, d = X.shape
        self.n_out = n + self.n_synth if self.order is None else self.order.shape[0]
        self.out = np.empty((self.n_out, d), dtype=np.float64)
        self.n_threads = int(n_threads) if n_threads else max(1, min(16, os.cpu_count() or 1))

    def _run(self, synth, which):
        n, d = self.X.shape
        rc = self._lib.smrt_assemble_rows_part(
            self.X.ctypes.data, int(self.X.dtype == np.float32), n, d, d, synth.ctypes.data if synth is not None else None,
            self.n_synth, d, self.order.ctypes.data if self.order is not None else None, self.n_out, self.out.ctypes.data,
            self.n_threads, which)
        if rc != 0:
            raise HipLibraryError("smrt_assemble_rows_part rejected its arguments (%d)" % rc)

    def originals(self):
        self._run(None, 1)

    def _check_synth(self, synth):
        if synth.dtype != np.float32 or not synth.flags.c_contiguous:
            synth = np.ascontiguousarray(synth, dtype=np.float32)
        if synth.shape != (self.n_synth, self.X.shape[1]):
            raise ValueError("expected %r synthetic rows, got %r" % ((self.n_synth, self.X.shape[1]), synth.shape))
        return synth

    def synthetic(self, synth):
        synth = self._check_synth(synth)
        self.wait()
        if self.n_synth:
            self._run(synth, 2)
        return self.out

    def synthetic_range_async(self, synth, lo, hi):
        """Start writing the output rows whose source is one of the synthetic rows ``[lo, hi)`` of the slab ``synth`` (the
        slab itself, not a copy: it must stay alive and those rows must be final) on a background thread; ``wait()``
        joins everything started so far.  ``originals()`` must have run (it validates ``order``)."""
        synth = self._check_synth(synth)
        if getattr(self, "_bg", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._bg, self._pending = ThreadPoolExecutor(max_workers=1), []
        n, d = self.X.shape

        def job():
            rc = self._lib.smrt_assemble_rows_synth_range(
                synth.ctypes.data, n, self.n_synth, d, d, self.order.ctypes.data if self.order is not None else None,
                self.n_out, self.out.ctypes.data, self.n_threads, int(lo), int(hi))
            if rc != 0:
                raise HipLibraryError("smrt_assemble_rows_synth_range rejected its arguments (%d)" % rc)

        self._pending.append(self._bg.submit(job))

    def wait(self):
        """Join the ranges started by ``synthetic_range_async`` (re-raises their errors); returns ``out``."""
        bg = getattr(self, "_bg", None)
        if bg is not None:
            try:
                for f in self._pending:
                    f.result()
            finally:
                bg.shutdown(wait=True)
                self._bg, self._pending = None, []
        return self.out


def _dev_f64(a, device):
    """Host array or device tensor -> contiguous fp64 [n, d] tensor on ``device``."""
    torch = _torch()
    if hasattr(a, "data_ptr"):
        t = a.to(device=device, dtype=torch.float64).contiguous()
    else:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    if t.dim() != 2:
        raise ValueError("expected a 2-d matrix")
    return t


def knn_f64(C_rows, k, Q_rows=None, device=0, return_distance=False, to_host=True):
    """The ``k`` rows of ``C_rows`` nearest (Euclidean, exact fp64, ties to the lower row) to each row of ``Q_rows``
    (default: ``C_rows`` itself, so column 0 is the row itself unless it has an exact duplicate).

    Returns int64 ``[nq, k]`` indices, preceded by the fp64 distances when ``return_distance``."""
    torch = _torch()
    dev = torch.device("cuda", int(device))
    Ct = _dev_f64(C_rows, dev)
    Qt = Ct if Q_rows is None else _dev_f64(Q_rows, dev)
    if Qt.shape[1] != Ct.shape[1]:
        raise ValueError("query rows have %d columns, fitted rows have %d" % (Qt.shape[1], Ct.shape[1]))
    nq, nc, d = Qt.shape[0], Ct.shape[0], Ct.shape[1]
    idx = torch.empty((nq, int(k)), dtype=torch.int64, device=dev)
    rd = torch.empty((nq, int(k)), dtype=torch.float64, device=dev) if return_distance else None
    stream = torch.cuda.current_stream(dev)
    _check(load().smrt_knn_f64(Qt.data_ptr(), nq, Qt.stride(0), Ct.data_ptr(), nc, Ct.stride(0), d, int(k), idx.data_ptr(),
                               rd.data_ptr() if rd is not None else None, C.c_void_p(stream.cuda_stream)), "smrt_knn_f64")
    if return_distance:
        rd = torch.sqrt(rd)                                  # the kernel ranks on squared distances
        return (rd.cpu().numpy(), idx.cpu().numpy()) if to_host else (rd, idx)
    return idx.cpu().numpy() if to_host else idx


def smote_rows(X_dev, base_idx, nbr_idx, r, mode="perturb", to_host=True):
    """Synthetic SMOTE rows from device-resident fp64 ``X_dev``: ``(x_b - x_n) * r + x_b`` or the midpoint."""
    torch = _torch()
    dev = X_dev.device
    base = torch.from_numpy(np.ascontiguousarray(base_idx, dtype=np.int64)).to(dev)
    nbr = torch.from_numpy(np.ascontiguousarray(nbr_idx, dtype=np.int64)).to(dev)
    n = base.shape[0]
    if nbr.shape[0] != n:
        raise ValueError("base and neighbour index vectors differ in length")
    rt = None
    if mode == "perturb":
        rt = torch.from_numpy(np.ascontiguousarray(r, dtype=np.float64)).to(dev)
        if rt.shape[0] != n:
            raise ValueError("one random factor per synthetic row is required")
    out = torch.empty((n, X_dev.shape[1]), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)
    _check(load().smrt_smote_rows_f64(X_dev.data_ptr(), X_dev.shape[0], X_dev.stride(0), X_dev.shape[1], base.data_ptr(),
                                      nbr.data_ptr(), rt.data_ptr() if rt is not None else None, n,
                                      0 if mode == "perturb" else 1, out.data_ptr(), out.stride(0),
                                      C.c_void_p(stream.cuda_stream)), "smrt_smote_rows_f64")
    return out.cpu().numpy() if to_host else out


def persistent_budget(device=0):
    """Persistent workgroups this process may keep resident on ``device`` (include/smrt_hip.h)."""
    return int(load().smrt_persistent_budget(int(device)))


def train_epochs_multi(handles, Xs, batch_size, n_epochs):
    """The ``_train`` loops of several handles side by side with grouped launches (``smrt_vae_train_epochs_multi``).

    ``Xs[k]`` is ``(tensor, ld)`` on the handles' GPU.  Returns ``(costs [n_handles][n_epochs], epoch_ms [n_epochs])``, or
    ``None`` when the library declines (nets on the LDS-resident paths, batches beyond 512 rows, a single handle): the
    caller then trains the handles one by one."""
    n = len(handles)
    lib = load()
    hs = (C.c_void_p * n)(*[h._live() for h in handles])
    xp = (C.c_void_p * n)(*[C.c_void_p(X[0].data_ptr()) for X in Xs])
    ld = (C.c_int64 * n)(*[int(X[1]) for X in Xs])
    nr = (C.c_int64 * n)(*[int(X[0].shape[0]) for X in Xs])
    costs = np.zeros((n, int(n_epochs)), dtype=np.float32)
    ms = np.zeros(int(n_epochs), dtype=np.float32)
    sync_torch_stream(handles[0].device)
    rc = lib.smrt_vae_train_epochs_multi(hs, n, xp, ld, nr, int(batch_size), int(n_epochs), _f32p(costs), _f32p(ms))
    if rc == 1:
        return None
    _check(rc, "smrt_vae_train_epochs_multi")
    return costs, ms


class VaeHandle(object):
    """Owns one ``smrt_vae*``.  Methods take/return numpy arrays or torch device tensors."""

    def __init__(self, n_features, hidden, n_latent, activation, optimizer, learning_rate, l2_penalty, clip,
                 gclip_min, gclip_max, eps, dropout, seed, max_batch, device=0, model="vae"):
        lib = load()
        torch = _torch()
        if len(hidden) > MAX_HIDDEN:
            raise ValueError("at most %d hidden layers are supported" % MAX_HIDDEN)
        cfg = VaeCfg()
        cfg.n_features, cfg.n_hidden_layers, cfg.n_latent = int(n_features), len(hidden), int(n_latent)
        for i, hdim in enumerate(hidden):
            cfg.hidden[i] = int(hdim)
        cfg.activation, cfg.optimizer = ACTIVATION_IDS[activation], OPTIMIZER_IDS[optimizer]
        cfg.learning_rate = float(learning_rate)
        cfg.l2_penalty = -1.0 if l2_penalty is None else float(l2_penalty)
        cfg.clip, cfg.gclip_min, cfg.gclip_max = int(bool(clip)), float(gclip_min), float(gclip_max)
        # clip bounds become fp32 constants of the graph: float32(1 - 1e-10) == 1.0f
        cfg.bce_lo, cfg.bce_hi = float(np.float32(eps)), float(np.float32(1.0 - eps))
        cfg.dropout_keep = float(dropout)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.max_batch = int(max_batch)
        cfg.model = {"vae": 0, "ae": 1}[model]
        self.model = model
        self.code_width = int(n_latent) if model == "vae" else int(hidden[-1])
        self.device = torch.device("cuda", int(device))
        self.n_features, self.n_latent, self.hidden = int(n_features), int(n_latent), list(hidden)
        self._h = C.c_void_p()
        _check(lib.smrt_vae_create(C.byref(cfg), int(device), C.byref(self._h)), "smrt_vae_create")
        self._lib = lib
        self.n_params = int(lib.smrt_vae_param_count(self._h))

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.smrt_vae_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def closed(self):
        return not self._h.value

    def _live(self):
        if self.closed:
            raise HipLibraryError("VAE handle is closed")
        return self._h

    # -- parameters / optimizer state (flat ABI order)
    def set_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        _check(self._lib.smrt_vae_set_params(self._live(), _f32p(flat), flat.size), "smrt_vae_set_params")

    def get_params(self):
        out = np.empty(self.n_params, dtype=np.float32)
        _check(self._lib.smrt_vae_get_params(self._live(), _f32p(out), out.size), "smrt_vae_get_params")
        return out

    def get_opt_state(self, slot):
        out = np.empty(self.n_params, dtype=np.float32)
        _check(self._lib.smrt_vae_get_opt_state(self._live(), int(slot), _f32p(out), out.size), "smrt_vae_get_opt_state")
        return out

    def reset_opt_state(self):
        _check(self._lib.smrt_vae_reset_opt_state(self._live()), "smrt_vae_reset_opt_state")

    def set_dropout_masks(self, flat_or_none):
        if flat_or_none is None:
            _check(self._lib.smrt_vae_set_dropout_masks(self._live(), None, 0), "smrt_vae_set_dropout_masks")
            return
        flat = np.ascontiguousarray(flat_or_none, dtype=np.float32)
        _check(self._lib.smrt_vae_set_dropout_masks(self._live(), _f32p(flat), flat.size), "smrt_vae_set_dropout_masks")

    @property
    def step_count(self):
        return int(self._lib.smrt_vae_step_count(self._live()))

    # -- training
    def _as_dev(self, X):
        """numpy [n,d], a contiguous fp32 device tensor [n, d % 4 == 0] or (tensor, ld) -> (tensor, ld) on this device."""
        if isinstance(X, tuple):
            return X
        if hasattr(X, "data_ptr"):
            # only an fp32 tensor that already sits on THIS GPU in the kernels' layout is passed through as a pointer;
            # anything else (CPU tensor, fp64, another device, ragged width) goes through the host conversion
            if (X.dim() == 2 and X.dtype == _torch().float32 and X.device == self.device and X.shape[1] % 4 == 0
                    and X.is_contiguous()):
                return X, X.shape[1]
            X = X.detach().cpu().numpy()
        return to_device_padded(X, self.device)

    def train_step(self, X, eps=None, want_grads=False):
        torch = _torch()
        Xt, ldx = self._as_dev(X)
        rows = Xt.shape[0]
        eps_ptr, eps_t = None, None
        if eps is not None:
            eps_t = torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32)).to(self.device)
            assert eps_t.shape == (rows, self.n_latent)
            eps_ptr = eps_t.data_ptr()
        sync_torch_stream(self.device)
        cost = C.c_float()
        grads = np.empty(self.n_params, dtype=np.float32) if want_grads else None
        _check(self._lib.smrt_vae_train_step(self._live(), Xt.data_ptr(), ldx, rows, eps_ptr, C.byref(cost),
                                             _f32p(grads) if want_grads else None), "smrt_vae_train_step")
        return (np.float32(cost.value), grads) if want_grads else np.float32(cost.value)

    def train_epochs(self, X, batch_size, n_epochs, early_stopping=False, min_change=1e-3):
        Xt, ldx = self._as_dev(X)
        sync_torch_stream(self.device)
        n = Xt.shape[0]
        costs = np.zeros(max(n_epochs, 1), dtype=np.float32)
        ms = np.zeros(max(n_epochs, 1), dtype=np.float32)
        done = C.c_int32()
        _check(self._lib.smrt_vae_train_epochs(self._live(), Xt.data_ptr(), ldx, n, int(batch_size), int(n_epochs),
                                               int(bool(early_stopping)), float(min_change), _f32p(costs), _f32p(ms),
                                               C.byref(done)), "smrt_vae_train_epochs")
        return costs[:done.value].copy(), ms[:done.value].copy()

    PROFILE_KINDS = ("dropout", "fwd_gemm", "fwd_gemm_bce", "reparam", "bwd_gemm", "bwd_gemm_latent", "dw_gemm",
                     "optimizer", "finalize", "reduce_slabs", "fused_fwd_bwd", "fused_dw_opt", "persist_train")

    def profile_epochs(self, X, batch_size, n_epochs):
        """Eager epochs with a hipEvent pair around every kernel: {kind: (total ms, launches)}."""
        Xt, ldx = self._as_dev(X)
        nk = len(self.PROFILE_KINDS)
        ms = (C.c_double * nk)()
        cnt = (C.c_int64 * nk)()
        _check(self._lib.smrt_vae_profile_epochs(self._live(), Xt.data_ptr(), ldx, Xt.shape[0], int(batch_size),
                                                 int(n_epochs), ms, cnt), "smrt_vae_profile_epochs")
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(self.PROFILE_KINDS)}

    def persistent_workgroups(self, batch_rows):
        """Workgroups one persistent whole-epoch launch of this handle holds (0: the net or the schedule does not qualify)."""
        return int(self._lib.smrt_vae_persistent_workgroups(self._live(), int(batch_rows)))

    @property
    def persistent_form(self):
        """0: not (yet) on the persistent path; 1: the generic persistent kernel; 2: one written out for this net's shape."""
        return int(self._lib.smrt_vae_persistent_form(self._live()))

    def set_persistent(self, mode):
        """-1: persistent path when the device's budget has room at call time; 0: never; 1: always (budget planned by the caller)."""
        _check(self._lib.smrt_vae_set_persistent(self._live(), int(mode)), "smrt_vae_set_persistent")

    @property
    def last_epoch_launches(self):
        return int(self._lib.smrt_vae_last_epoch_launches(self._live()))

    # -- data-parallel halves
    @property
    def dp_count(self):
        return int(self._lib.smrt_vae_dp_count(self._live()))

    def grad_step(self, X, global_rows, row0_global, grads_tensor, eps=None):
        torch = _torch()
        Xt, ldx = self._as_dev(X)
        eps_ptr, eps_t = None, None
        if eps is not None:
            eps_t = torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32)).to(self.device)
            eps_ptr = eps_t.data_ptr()
        sync_torch_stream(self.device)
        _check(self._lib.smrt_vae_grad_step(self._live(), Xt.data_ptr(), ldx, Xt.shape[0], int(global_rows),
                                            int(row0_global), eps_ptr, grads_tensor.data_ptr()), "smrt_vae_grad_step")

    def apply_grads(self, grads_tensor):
        sync_torch_stream(self.device)
        cost = C.c_float()
        _check(self._lib.smrt_vae_apply_grads(self._live(), grads_tensor.data_ptr(), C.byref(cost)), "smrt_vae_apply_grads")
        return np.float32(cost.value)

    def use_stream(self, torch_stream=None):
        """Enqueue on ``torch_stream`` (a ``torch.cuda.Stream``) from now on; ``None`` returns to the private stream."""
        if torch_stream is None:
            _check(self._lib.smrt_vae_set_stream(self._live(), None, 1), "smrt_vae_set_stream")
        else:
            _check(self._lib.smrt_vae_set_stream(self._live(), C.c_void_p(torch_stream.cuda_stream), 0),
                   "smrt_vae_set_stream")

    def grad_step_async(self, X_ptr, ldx, rows, global_rows, row0_global, grads_tensor):
        """Raw-pointer form for host loops: no synchronisation, Philox epsilon by global row."""
        _check(self._lib.smrt_vae_grad_step_async(self._live(), X_ptr, int(ldx), int(rows), int(global_rows),
                                                  int(row0_global), None, grads_tensor.data_ptr()),
               "smrt_vae_grad_step_async")

    def apply_grads_async(self, grads_tensor, cost_tensor=None):
        _check(self._lib.smrt_vae_apply_grads_async(self._live(), grads_tensor.data_ptr(),
                                                    None if cost_tensor is None else cost_tensor.data_ptr()),
               "smrt_vae_apply_grads_async")

    def dp_grad_step(self, X_local, row_offset, rows, global_rows, row0_global, grads_tensor):
        """Stripe ``[row_offset, row_offset + rows)`` of the rank-local matrix ``X_local = (tensor, ld)``."""
        Xt, ldx = X_local
        self.grad_step_async(Xt.data_ptr() + 4 * int(row_offset) * int(ldx), ldx, rows, global_rows, row0_global,
                             grads_tensor)

    def dp_apply(self, grads_tensor, cost_tensor=None):
        self.apply_grads_async(grads_tensor, cost_tensor)

    # -- inference
    def encode(self, X, to_host=True):
        torch = _torch()
        Xt, ldx = self._as_dev(X)
        n = Xt.shape[0]
        mu = torch.empty((n, self.code_width), dtype=torch.float32, device=self.device)
        if self.model == "ae":                          # the code of the plain auto-encoder: one array
            sync_torch_stream(self.device)
            if n:
                _check(self._lib.smrt_vae_encode(self._live(), Xt.data_ptr(), ldx, n, mu.data_ptr(), None), "smrt_vae_encode")
            return mu.cpu().numpy() if to_host else mu
        ls = torch.empty((n, self.n_latent), dtype=torch.float32, device=self.device)
        sync_torch_stream(self.device)
        if n:
            _check(self._lib.smrt_vae_encode(self._live(), Xt.data_ptr(), ldx, n, mu.data_ptr(), ls.data_ptr()),
                   "smrt_vae_encode")
        return (mu.cpu().numpy(), ls.cpu().numpy()) if to_host else (mu, ls)

    def decode(self, Z, to_host=True):
        torch = _torch()
        Zt, ldz = self._as_dev(Z)
        n = Zt.shape[0]
        out = torch.empty((n, self.n_features), dtype=torch.float32, device=self.device) \
            if self.n_features % 4 == 0 else torch.empty((n, rup4(self.n_features)), dtype=torch.float32, device=self.device)
        sync_torch_stream(self.device)
        if n:
            _check(self._lib.smrt_vae_decode(self._live(), Zt.data_ptr(), ldz, n, out.data_ptr(), out.shape[1]),
                   "smrt_vae_decode")
        out = out[:, :self.n_features]
        return out.cpu().numpy() if to_host else out

    def generate(self, n, philox_offset=0, loc=0.0, scale=1.0, to_host=True):
        torch = _torch()
        out = torch.empty((n, rup4(self.n_features)), dtype=torch.float32, device=self.device)
        sync_torch_stream(self.device)
        if n:
            _check(self._lib.smrt_vae_generate(self._live(), n, int(philox_offset), float(loc), float(scale),
                                               out.data_ptr(), out.shape[1]), "smrt_vae_generate")
        out = out[:, :self.n_features]
        return out.cpu().numpy() if to_host else out

    @property
    def generate_noalloc_max_rows(self):
        return int(self._lib.smrt_vae_generate_noalloc_max_rows(self._live()))

    def generate_noalloc(self, out, n, philox_offset=0, loc=0.0, scale=1.0):
        """``generate`` into the caller's fp32 device tensor ``out`` [>= n, ld >= n_features] through the entry point
        that allocates nothing (static per-GPU scratch, ``generate_noalloc_max_rows`` rows at most)."""
        _check(self._lib.smrt_vae_generate_noalloc(self._live(), int(n), int(philox_offset), float(loc), float(scale),
                                                   out.data_ptr(), out.stride(0)), "smrt_vae_generate_noalloc")

    def sample_eps(self, step, row0, rows):
        torch = _torch()
        out = torch.empty((rows, self.n_latent), dtype=torch.float32, device=self.device)
        sync_torch_stream(self.device)
        _check(self._lib.smrt_vae_sample_eps(self._live(), int(step), int(row0), int(rows), out.data_ptr()),
               "smrt_vae_sample_eps")
        return out.cpu().numpy()
