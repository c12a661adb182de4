"""Data-parallel training of ONE variational auto-encoder across ranks (one process per GPU).

No reference counterpart: the reference trains on one CPU (smrt/autoencode/base.py:183-254).  The schedule
it fixes is kept exactly -- contiguous unshuffled mini-batches with a ragged tail (``gen_batches``,
base.py:198-199), cost of each epoch's LAST batch, early stopping on ``|last - c| <= min_change`` -- and every
mini-batch ``[s, s + b)`` is cut into ``world`` contiguous stripes, one per rank:

    rank r trains rows  [s + r*b // world, s + (r+1)*b // world)

Per step each rank leaves ``[gradient sums | loss sum]`` scaled by ``1/b`` in one flat device buffer
(``smrt_vae_grad_step_async``), the ranks ``all_reduce(SUM)`` it (RCCL over xGMI when the backend is ``nccl``),
and every rank applies L2, the value clip and the optimizer to its replica (``smrt_vae_apply_grads_async``):
the clip comes after the reduction and L2 is added once, so the update equals the single-GPU update up to
fp32 summation order; the reparameterisation epsilon is indexed by (step, row within the GLOBAL batch),
so it is the same draw.
Nothing in the step synchronises with the host: the handle is lent the stream the collective is ordered
against, and the cost is read back once per epoch.
"""
from __future__ import annotations

import time

import numpy as np

__all__ = ["plan_stripes", "local_row_index", "DataParallelTrainer", "dist_info"]


def dist_info():
    """(rank, world, module) when torch.distributed is initialised, else (0, 1, None)."""
    try:
        import torch.distributed as dist
    except Exception:       # pragma: no cover
        return 0, 1, None
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist
    return 0, 1, None


def plan_stripes(n_rows, batch_size, world):
    """One entry per mini-batch, in training order: ``(start, rows, bounds)`` where ``bounds`` holds the
    ``world + 1`` stripe boundaries (global row numbers) of that batch."""
    n_rows, batch_size, world = int(n_rows), int(batch_size), int(world)
    if n_rows < 1 or batch_size < 1 or world < 1:
        raise ValueError("n_rows, batch_size and world must be >= 1")
    plan = []
    for start in range(0, n_rows, batch_size):
        b = min(batch_size, n_rows - start)
        bounds = start + (np.arange(world + 1, dtype=np.int64) * b) // world
        plan.append((start, b, bounds))
    return plan


def local_row_index(n_rows, batch_size, world, rank):
    """Global row numbers held by ``rank``: its stripe of every batch, concatenated in batch order."""
    parts = [np.arange(bounds[rank], bounds[rank + 1], dtype=np.int64)
             for _, _, bounds in plan_stripes(n_rows, batch_size, world)]
    return np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)


class DataParallelTrainer(object):
    """Runs the reference's ``_train`` loop over row stripes that already sit in this rank's HBM.

    ``sess`` is a ``smrt_amd._hip.VaeHandle`` (or anything with ``device``, ``dp_count``, ``use_stream``,
    ``dp_grad_step`` and ``dp_apply``); ``X_local`` is ``(tensor [n_local, ld], ld)`` holding
    ``local_row_index(...)`` rows in that order.
    """

    def __init__(self, sess, X_local, n_rows, batch_size, rank, world, dist):
        import torch
        self.sess, self.dist, self.rank, self.world = sess, dist, int(rank), int(world)
        self.X_local = X_local
        self.plan = plan_stripes(n_rows, batch_size, world)
        n_local = sum(int(b[rank + 1] - b[rank]) for _, _, b in self.plan)
        if int(X_local[0].shape[0]) != n_local:
            raise ValueError("rank %d holds %d rows, its stripes need %d" % (rank, X_local[0].shape[0], n_local))
        self.grads = torch.zeros(sess.dp_count, dtype=torch.float32, device=sess.device)
        self.on_gpu = sess.device.type == "cuda"
        # A host-side backend (gloo: the one-GPU rehearsal of this loop) reduces HOST tensors here: the gradient buffer is
        # copied down (a stream-ordered copy that waits for exactly the kernels that wrote it), reduced, and copied back.
        # gloo is never handed a device tensor.  Round 1 did hand it one and found that a gloo all_reduce of a device
        # tensor entered while kernels were still queued on the current (legacy) stream never returned on this image; every
        # launch of the two async halves is on the lent stream (api.hip: enqueue_step / enqueue_apply / the rows == 0
        # memset all take h->stream), so the wait was inside gloo's own device-tensor path (its worker thread's private
        # stream against the legacy stream), which this loop no longer enters.  RCCL is ordered on the stream itself.
        self.host_collective = bool(self.on_gpu and dist is not None and self.world > 1 and dist.get_backend() != "nccl")
        # measurement aid (bench.py: dp_c4): with `time_collective` set, every RCCL all-reduce is bracketed by a pair of
        # events on the stream it is ordered on (no host wait); `collective_ms` then holds the per-step device times of
        # the most recent run_epochs call.  Nothing overlaps the collective in this loop, so all of it is exposed.
        self.time_collective = False
        self.collective_ms = []

    def run_epochs(self, n_epochs, early_stopping=False, min_change=1e-3):
        """Returns ``(costs, seconds)`` per finished epoch (cost of the last batch, base.py:222)."""
        import torch
        sess, dist, rank = self.sess, self.dist, self.rank
        cost = torch.zeros(max(int(n_epochs), 1), dtype=torch.float32, device=sess.device)
        costs, seconds, last = [], [], None
        stream = torch.cuda.current_stream(sess.device) if self.on_gpu else None
        sess.use_stream(stream)
        timed = []
        try:
            for ep in range(int(n_epochs)):
                if self.on_gpu:
                    torch.cuda.synchronize(sess.device)
                t0 = time.perf_counter()
                off = 0
                for i, (start, b, bounds) in enumerate(self.plan):
                    rows = int(bounds[rank + 1] - bounds[rank])
                    sess.dp_grad_step(self.X_local, off, rows, b, int(bounds[rank] - start), self.grads)
                    off += rows
                    if dist is not None and self.world > 1:
                        if self.host_collective:
                            g_host = self.grads.cpu()
                            dist.all_reduce(g_host)
                            self.grads.copy_(g_host)
                        elif self.time_collective and self.on_gpu:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(stream)
                            dist.all_reduce(self.grads)
                            e1.record(stream)
                            timed.append((e0, e1))
                        else:
                            dist.all_reduce(self.grads)                # SUM; RCCL ring / tree over xGMI under nccl
                    sess.dp_apply(self.grads, cost[ep:ep + 1] if i == len(self.plan) - 1 else None)
                if early_stopping and dist is not None and self.world > 1:     # one decision for all ranks
                    if self.host_collective:
                        c_host = cost[ep:ep + 1].cpu()
                        dist.broadcast(c_host, src=0)
                        cost[ep:ep + 1].copy_(c_host)
                    else:
                        dist.broadcast(cost[ep:ep + 1], src=0)
                c = float(cost[ep].item())                             # the epoch's only host synchronisation
                seconds.append(time.perf_counter() - t0)
                costs.append(np.float32(c))
                if early_stopping:                                     # base.py:230-240
                    if last is None:
                        last = c
                    elif abs(last - c) <= min_change:
                        break
                    else:
                        last = c
        finally:
            sess.use_stream(None)
        if timed:
            torch.cuda.synchronize(sess.device)
            self.collective_ms = [e0.elapsed_time(e1) for e0, e1 in timed]
        return np.asarray(costs, dtype=np.float32), seconds
