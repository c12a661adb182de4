"""``smrt_balance`` -- per-minority-class VAE fit + synthesis on MI355X.

Drop-in for the reference function (smrt/balance/smrt.py:22-250): same signature, same return
values, same consumption order of the ONE shared numpy stream (per class: ``choice`` when
``gen_from_samples``, then the latent ``normal`` draw; finally the shuffle ``permutation``).

What is different under the hood:
  * every VAE trains and generates through the HIP library (``smrt_amd.autoencode``);
  * the balanced matrix is written ONCE: the input is validated in place (no fp64 copy up front), the
    synthetic blocks are collected in one fp32 slab, and ``smrt_assemble_rows`` (multi-threaded C++, host
    side of the library) writes ``fp64([X ; synthetic][order])`` in a single pass -- the reference makes two
    full copies of X, re-copies the whole matrix per class and gathers it again for the shuffle
    (base.py:24, smrt.py:184,237,248-250);
  * under ``torch.distributed`` (one process per GPU) the independent per-class jobs are sharded
    across ranks, longest job first, with no collective on the data path; the finished blocks are
    exchanged once at the end.  The result is identical for any world size because ranks that do
    not own a class still advance the shared stream by that class's draws.
  * the reference masks the original ``X`` with a label vector it has already extended
    (smrt.py:207,238), which numpy >= 1.13 rejects once a second class needs synthesis; the mask
    here is over the original labels (what numpy 1.12 computed).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from .. import _hip
from ..autoencode import VariationalAutoEncoder
from ..autoencode.topography import resolve_latent
from ..autoencode.vae import fit_group
from ..utils import DEFAULT_SEED, get_random_state
from .base import validate_X_y_ratio_classes

__all__ = ["smrt_balance", "plan_class_jobs"]


def _dist():
    """(rank, world, module) when torch.distributed is initialised, else (0, 1, None)."""
    try:
        import torch.distributed as dist
    except Exception:       # pragma: no cover
        return 0, 1, None
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist
    return 0, 1, None


def plan_class_jobs(costs, world):
    """Greedy longest-job-first assignment of jobs to ``world`` ranks; returns owner per job.

    Deterministic (ties broken by job order, then by lowest rank) so every rank computes the same plan.
    """
    owners = [0] * len(costs)
    load = [0.0] * world
    for j in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(world), key=lambda q: (load[q], q))
        owners[j] = r
        load[r] += costs[j]
    return owners


def _exchange_block(dist, slab_rows, src, mine, rows_dev=None):
    """Broadcast one finished fp32 block from its owner to every rank, in place in ``slab_rows`` (this rank's view of
    the block inside the synthetic slab; the owner's view already holds the rows).

    RCCL (``nccl``) moves device memory: the owner sends the decoded rows straight from HBM when they are still there
    (``rows_dev``), every other rank receives into a device buffer and downloads it into its slab -- no staging copy
    on either side.  A host backend (``gloo``) broadcasts the slab view itself."""
    import torch
    view = torch.from_numpy(slab_rows)
    if dist.get_backend() != "nccl":
        dist.broadcast(view, src=src)
        return
    dev = torch.device("cuda", torch.cuda.current_device())
    if mine:
        t = rows_dev if rows_dev is not None else view.to(dev)
        if not t.is_contiguous() or t.dtype != torch.float32:
            t = t.contiguous().float()
    else:
        t = torch.empty(tuple(slab_rows.shape), dtype=torch.float32, device=dev)
    dist.broadcast(t, src=src)
    if not mine:
        view.copy_(t)


class _GroupMember(object):
    """``result()`` of one estimator of a ``fit_group`` task (the interface of a future)."""

    def __init__(self, group_future, estimator):
        self.group_future, self.estimator = group_future, estimator

    def result(self):
        self.group_future.result()
        return self.estimator


def smrt_balance(X, y, n_hidden, n_latent_factors, return_estimators=False, balance_ratio=0.2,
                 activation_function='sigmoid', learning_rate=0.05, n_epochs=20, batch_size=128, min_change=1e-3,
                 verbose=0, display_step=5, learning_function='rms_prop', early_stopping=False, bias_strategy='zeros',
                 random_state=DEFAULT_SEED, layer_type='xavier', dropout=1., l2_penalty=0.0001,
                 eps=1e-10, gclip_min=-5., gclip_max=5., clip=True, shuffle=True, gen_from_samples=False,
                 generate_args=None, prefit_estimators=None):
    """Balance ``(X, y)`` by synthesising minority rows with one variational auto-encoder per class.

    Every class that is not the majority and has fewer than
    ``target = max(int(balance_ratio * majority_count), 1)`` rows gets a ``VariationalAutoEncoder`` fitted on its
    rows (or taken from ``prefit_estimators[label]``) and ``target - count`` synthetic rows from
    ``generate`` (or, with ``gen_from_samples``, ``generate_from_sample`` on rows re-drawn with replacement).
    Remaining keyword arguments are the ``VariationalAutoEncoder`` constructor arguments;
    ``generate_args`` (``loc`` / ``scale``) is forwarded to the latent normal draw.

    Returns ``(X_balanced, y_balanced)`` -- fp64 rows, labels in the caller's label space, shuffled by the
    shared random state unless ``shuffle=False`` -- plus ``{label: estimator or None}`` when
    ``return_estimators``.
    """
    timing = os.environ.get("SMRT_TIMING") == "1"
    t_phase = [time.perf_counter()]

    def lap(what):
        if timing:
            t_phase.append(time.perf_counter())
            sys.stderr.write("[smrt_balance] %-10s %8.1f ms\n" % (what, 1e3 * (t_phase[-1] - t_phase[-2])))

    X, y, n_classes, present_classes, counts, majority_label, target_count = \
        validate_X_y_ratio_classes(X, y, balance_ratio, copy=False, keep_float32=True)
    lap("validate")

    seeded_random_state = get_random_state(random_state)
    random_state = seeded_random_state.state

    if present_classes.dtype.kind in "iu" and 0 <= int(present_classes[0]) and int(present_classes[-1]) < (1 << 16):
        lut = np.zeros(int(present_classes[-1]) + 1, dtype=np.intp)
        lut[present_classes] = np.arange(len(present_classes))
        y_codes = lut[y]                                  # same codes as below, one gather instead of a binary search per row
    else:
        y_codes = np.searchsorted(present_classes, y)     # == LabelEncoder().fit(present_classes).transform(y)

    if generate_args is None:
        generate_args = dict()
    if prefit_estimators is None:
        prefit_estimators = dict()
    elif not isinstance(prefit_estimators, dict):
        raise ValueError('if prefit estimators are provided, it must be a dict')

    # ---- plan: which classes need rows, how many, who owns each job
    n_orig, n_features = X.shape
    jobs = []                                             # (class code, label, rows in class, rows needed)
    encoders = dict()
    for code, label in enumerate(present_classes):
        if label == majority_label or counts[code] >= target_count:
            encoders[label] = None
            continue
        encoders[label] = None
        jobs.append((code, label, int(counts[code]), int(target_count - counts[code])))

    rank, world, dist = _dist()
    owners = plan_class_jobs([n_sub * max(int(n_epochs), 1) + need for _, _, n_sub, need in jobs], world)

    n_synth = sum(j[3] for j in jobs)
    n_total = n_orig + n_synth
    # what the kernels produce (fp32, autoencoder.py:517).  The slab is page-locked when a GPU fills it (torch's caching host
    # allocator: free after the first call): the decoded rows of C3 are 502 MB of downloads on the critical path, 33 ms into
    # pageable memory, 14 ms into pinned.  SMRT_PIN_SLAB_MB caps it (default 4096, 0: never).
    synth = None
    pinned = False
    pin_cap = int(os.environ.get("SMRT_PIN_SLAB_MB", "4096")) << 20
    if 0 < n_synth * n_features * 4 <= pin_cap:
        import torch
        if torch.cuda.is_available():
            try:
                synth = torch.empty((n_synth, n_features), dtype=torch.float32, pin_memory=True).numpy()
                pinned = True
            except RuntimeError:
                synth = None
    if synth is None:
        synth = np.empty((n_synth, n_features), dtype=np.float32)
    y_out_codes = np.empty(n_total, dtype=y_codes.dtype)
    y_out_codes[:n_orig] = y_codes

    # ---- per-class train (smrt.py:197-223).  A fit draws nothing from the shared numpy stream (it only reads the seed),
    # and every host draw of the generation step has a size known up front, so three things overlap:
    #   * the fits of the classes this rank owns, one host thread and one HIP stream each (ctypes releases the GIL; a
    #     default-batch step keeps 8-120 of the 256 CUs busy, so several fit in one GPU side by side);
    #   * the main thread drawing, in class order, exactly what the reference draws between its fits (the re-draw
    #     indices, the latent normals) and, last, the output permutation -- the sequential Mersenne-Twister work that
    #     otherwise sits between the GPU phases;
    #   * nothing else: decoding waits for the fit it belongs to.
    import torch
    device = torch.cuda.current_device() if torch.cuda.is_available() else 0      # worker threads do not inherit it
    subs, fits, pool = [], [], None
    n_threads = max(1, int(os.environ.get("SMRT_CLASS_THREADS", "8")))
    # the rows of every class that needs a fit (smrt.py:199: X[y == label, :]) -- row gathers of a few MB each, taken side by
    # side when there are several (numpy's take releases the GIL: C3's eight 15 MB gathers 10-17 ms -> 3-4 ms)
    row_sets = [np.flatnonzero(y_codes == code) for code, _, _, _ in jobs]
    if len(jobs) > 1 and sum(len(r) for r in row_sets) * n_features * X.dtype.itemsize > (8 << 20):
        from concurrent.futures import ThreadPoolExecutor as _SplitPool
        with _SplitPool(max_workers=min(8, len(jobs))) as sp:
            subs = list(sp.map(lambda r: X.take(r, axis=0), row_sets))
    else:
        subs = [X.take(r, axis=0) for r in row_sets]
    for j, (code, label, n_sub, need) in enumerate(jobs):
        if owners[j] != rank or prefit_estimators.get(label, None) is not None:
            fits.append(None)
            continue
        est = VariationalAutoEncoder(
            n_hidden=n_hidden, n_latent_factors=n_latent_factors, activation_function=activation_function,
            learning_rate=learning_rate, n_epochs=n_epochs, batch_size=batch_size, min_change=min_change,
            verbose=verbose, display_step=display_step, learning_function=learning_function,
            early_stopping=early_stopping, bias_strategy=bias_strategy, random_state=seeded_random_state,
            layer_type=layer_type, dropout=dropout, l2_penalty=l2_penalty, eps=eps, gclip_min=gclip_min,
            gclip_max=gclip_max, clip=clip)
        est.device = device
        fits.append(est)
    n_fresh = sum(f is not None for f in fits)
    lap("split")

    blocks, cursor, drawn = [], 0, []
    assembled = set()                        # jobs whose synthetic rows are in the output already (synthetic_range_async)
    assembly = None
    dev_rows = dict()                        # job -> decoded rows still in HBM (sent from there under RCCL)
    try:
        if n_fresh:
            from concurrent.futures import ThreadPoolExecutor
            # ONE background task fits every class this rank owns: grouped launches where the library can group them
            # (fit_group), one host thread and HIP stream per class otherwise; the main thread goes on with the draws
            pool = ThreadPoolExecutor(max_workers=1)
            fresh_jobs = [j for j, f in enumerate(fits) if f is not None]
            group = pool.submit(fit_group, [fits[j] for j in fresh_jobs], [subs[j] for j in fresh_jobs], n_threads)
            fits = [None if f is None else _GroupMember(group, f) for f in fits]

        if not pinned:
            synth.fill(0.0)                  # touch the slab's pages while the fits run, not while they are waited for
        # ---- the shared stream, in class order (smrt.py:226-238 via autoencoder.py:519-572)
        for j, (code, label, n_sub, need) in enumerate(jobs):
            mine = owners[j] == rank
            X_sub = subs[j]
            encoder = prefit_estimators.get(label, None)
            fresh = encoder is None
            noise = sample_idcs = None
            if fresh:
                if gen_from_samples:
                    sample_idcs = random_state.choice(np.arange(n_sub), size=need, replace=True)
                z_dim = resolve_latent(n_latent_factors, n_features)
                noise = _hip.rng_normal(random_state, (need, z_dim), **generate_args)     # == random_state.normal(...), bit for bit
                if not mine:
                    noise = sample_idcs = None
            elif mine:
                # a caller-supplied estimator draws for itself, here, at its place in the order
                encoders[label] = encoder
                if not gen_from_samples:
                    synthetic = encoder.generate(n=need, **generate_args)
                else:
                    sample_idcs = random_state.choice(np.arange(n_sub), size=need, replace=True)
                    synthetic = encoder.generate_from_sample(X_sub[sample_idcs, :], **generate_args)
                synth[cursor:cursor + need] = np.asarray(synthetic)
            else:
                # another rank owns this class: consume exactly the draws it makes on the shared stream
                if gen_from_samples:
                    random_state.choice(np.arange(n_sub), size=need, replace=True)
                if getattr(encoder, "random_state_", None) is seeded_random_state:
                    _hip.rng_normal(random_state, (need, int(encoder.n_latent_factors)), **generate_args)
            drawn.append((noise, sample_idcs))
            blocks.append((cursor, need, owners[j], mine))
            y_out_codes[n_orig + cursor:n_orig + cursor + need] = code
            cursor += need
        output_order = _hip.rng_permutation(random_state, n_total) if shuffle else None     # == permutation(np.arange(n_total))
        lap("draws")
        # the rows that come from X are written now, while the GPU is still training (85 % of the output on C2)
        assembly = _hip.SplitAssembly(X, n_synth, output_order)
        assembly.originals()
        if not gen_from_samples:
            # latent draws to the kernels' dtype (what feed_forward does, autoencoder.py:517), still under the cover of the fits
            drawn = [(None if nz is None else nz.astype(np.float32), si) for nz, si in drawn]
        lap("originals")
        if pinned and n_fresh:
            # A page-locked slab's first CPU touch is a fault into the driver per page (502 MB: 160 ms on one thread, 57 ms
            # inside the threaded assembly): taken here, on several threads, while the GPU trains and the host has nothing
            # else to do.  (A slab that torch's allocator hands out again is mapped already: 1-2 ms.)
            from concurrent.futures import ThreadPoolExecutor as _TP
            n_touch = max(1, min(8, (os.cpu_count() or 8), n_synth // 4096))
            step = -(-n_synth // n_touch)
            with _TP(max_workers=n_touch) as tp:
                list(tp.map(lambda i: synth[i * step:(i + 1) * step].fill(0.0), range(n_touch)))
            lap("touch")

        # ---- per-class generate for the fresh fits: decode what was drawn above
        for j, (code, label, n_sub, need) in enumerate(jobs):
            if fits[j] is None:
                continue
            encoder = fits[j].result()
            encoders[label] = encoder
            noise, sample_idcs = drawn[j]
            start = blocks[j][0]
            if not gen_from_samples:
                # == generate(n=need, **generate_args); the decoded rows are downloaded straight into their slab
                rows_dev = encoder.sess.decode(noise, to_host=False)
                torch.from_numpy(synth[start:start + need]).copy_(rows_dev)
                if world > 1 and tuple(rows_dev.shape) == (need, n_features):
                    dev_rows[j] = rows_dev
                if world == 1 and len(jobs) > 1:
                    # their place in the output while the next class is decoded and downloaded (the download is a DMA into
                    # page-locked memory: the host's cores are free for this)
                    assembly.synthetic_range_async(synth, start, start + need)
                    assembled.add(j)
            else:
                z_mu, log_sigma = encoder.encode(subs[j][sample_idcs, :])         # == generate_from_sample(...)
                z_mu += noise * np.exp(log_sigma)
                synth[start:start + need] = np.asarray(encoder.feed_forward(z_mu))
            drawn[j] = None
    except BaseException:
        if assembly is not None:
            try:
                assembly.wait()              # no background writer may outlive the slab
            except Exception:                # noqa: BLE001  (the first error is the one to report)
                pass
        raise
    finally:
        if pool is not None:
            pool.shutdown(wait=True)

    lap("fit+gen")
    if world > 1:
        for j, (start, need, owner, mine) in enumerate(blocks):
            _exchange_block(dist, synth[start:start + need], owner, mine, dev_rows.pop(j, None))

    y_out = present_classes[y_out_codes]                  # back to the caller's label space (smrt.py:241)

    if shuffle:
        y_out = y_out[output_order]
    if assembled and len(assembled) == len(jobs):
        X_out = assembly.wait()                           # every class's rows went in behind its download
    else:
        assembly.wait()
        X_out = assembly.synthetic(synth)                 # the other rows of fp64 [X ; synthetic][order]
    lap("assemble")

    if return_estimators:
        return X_out, y_out, encoders
    return X_out, y_out
