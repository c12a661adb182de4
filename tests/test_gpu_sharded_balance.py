"""smrt_balance with FRESH fits sharded over ranks (reference loop smrt/balance/smrt.py:197-238), on the GPU.

tests/test_cpu_lib_and_dist.py shards prefit fake estimators on the CPU; what an 8-GPU C3 run actually executes -- a
fresh VariationalAutoEncoder fitted on the owning rank only, decoded there, the block handed to the other ranks
(`_exchange_block`), the ranks that do not own a class still consuming its draws -- runs here as two gloo ranks that share
cuda:0.  (The RCCL arm of `_exchange_block` needs one GPU per rank and cannot run on a one-GPU box.)

Bar: bit-identical to the one-rank call.  That holds because a class trains the same kernels on the same numbers
wherever it runs: per-layer nets go through the grouped launches (bit-identical to separate fits,
test_gpu_api.py), LDS-resident nets ALWAYS take the persistent launch -- the budget of co-resident workgroups (halved
here: two ranks on one GPU) only decides how many run at a time, never which step a class trains with.
"""
import os
import sys

import numpy as np
import pytest

from tests.test_data_parallel import _run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

# two calls per run: an LDS-resident net (persistent launch per class) and a net on the per-layer kernels (grouped launches)
CALLS = (dict(n_hidden=[128], n_latent_factors=16, n_epochs=4, batch_size=128),
         dict(n_hidden=[300], n_latent_factors=7, n_epochs=3, batch_size=96, learning_function="adam", learning_rate=0.002))
COMMON = dict(balance_ratio=0.6, random_state=13, return_estimators=True)


def _data(call):
    """6 classes: a majority of 900 rows, four minority classes that need rows (fresh fits), one already above the target."""
    rs = np.random.RandomState(31 + call)
    D = 64 if call == 0 else 200
    sizes = [900, 260, 333, 150, 700, 410]
    Xs, ys = [], []
    for label, n in enumerate(sizes):
        mu = rs.normal(size=D)
        Xs.append(1.0 / (1.0 + np.exp(-(mu + rs.normal(size=(n, D))))))
        ys.append(np.full(n, 10 * label + 3))
    X, y = np.vstack(Xs).astype(np.float32), np.concatenate(ys)
    p = rs.permutation(len(y))
    return X[p], y[p]


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
os.environ["LOCAL_WORLD_SIZE"] = sys.argv[2]          # what a launcher exports: the ranks share this box's one GPU
import torch, torch.distributed as dist
from smrt_amd.balance import smrt_balance
from tests.test_gpu_sharded_balance import CALLS, COMMON, _data

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
res = dict()
for c, kw in enumerate(CALLS):
    for gfs in (False, True):
        X, y = _data(c)
        Xb, yb, enc = smrt_balance(X, y, gen_from_samples=gfs, **dict(COMMON, **kw))
        tag = "c%d_g%d_" % (c, int(gfs))
        res[tag + "X"], res[tag + "y"] = Xb, yb
        res[tag + "owned"] = np.array(sorted(k for k, e in enc.items() if e is not None), dtype=np.int64)
        res[tag + "costs"] = np.array([enc[k].epoch_costs_ for k in sorted(enc) if enc[k] is not None], dtype=np.float64)
        res[tag + "launches"] = np.array([enc[k].sess.last_epoch_launches for k in sorted(enc) if enc[k] is not None])
        for e in enc.values():
            if e is not None:
                e.clean_session()
np.savez(os.path.join(out, "out%d.npz" % rank), **res)
dist.barrier()
dist.destroy_process_group()
"""


def test_fresh_fits_sharded_over_two_ranks_equal_the_single_rank_call(tmp_path, monkeypatch):
    from smrt_amd.balance import smrt_balance
    monkeypatch.setenv("SMRT_HIP_FUSED", "1")
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1")
    ref = dict()
    for c, kw in enumerate(CALLS):
        for gfs in (False, True):
            X, y = _data(c)
            Xb, yb, enc = smrt_balance(X, y, gen_from_samples=gfs, **dict(COMMON, **kw))
            owned = sorted(k for k, e in enc.items() if e is not None)
            assert owned == [13, 23, 33, 53], owned                       # 43 is already above the target, 3 is the majority
            assert np.unique(yb, return_counts=True)[1].tolist() == [900, 540, 540, 540, 700, 540]
            launches = [enc[k].sess.last_epoch_launches for k in owned]
            if c == 0:
                assert launches == [2] * 4, launches                      # every class: ONE persistent launch per epoch (+ the cost)
            ref[(c, gfs)] = (Xb, yb, np.array([enc[k].epoch_costs_ for k in owned], dtype=np.float64))
            for e in enc.values():
                if e is not None:
                    e.clean_session()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    _run_ranks(script, tmp_path, timeout=600)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(2)]
    for c in range(len(CALLS)):
        for gfs in (False, True):
            tag = "c%d_g%d_" % (c, int(gfs))
            Xb, yb, costs = ref[(c, gfs)]
            owned = [outs[r][tag + "owned"].tolist() for r in range(2)]
            # each class was trained on exactly one rank, and both ranks worked
            assert sorted(owned[0] + owned[1]) == [13, 23, 33, 53] and owned[0] and owned[1], owned
            for r in range(2):
                np.testing.assert_array_equal(outs[r][tag + "y"], yb)
                np.testing.assert_array_equal(outs[r][tag + "X"], Xb, err_msg="rank %d, call %d, gen_from_samples %s" % (r, c, gfs))
                if c == 0:
                    assert outs[r][tag + "launches"].tolist() == [2] * len(owned[r])
            both = {k: row for r in range(2) for k, row in zip(owned[r], outs[r][tag + "costs"])}
            np.testing.assert_array_equal(np.array([both[k] for k in (13, 23, 33, 53)]), costs)


def test_more_lds_resident_classes_than_the_budget_holds_is_deterministic(monkeypatch):
    """with more concurrent LDS-resident fits than co-resident persistent workgroups, which classes fell back to
    the two-launch step used to depend on thread timing (and the two steps agree to 2e-5 only).  Now every class takes
    the persistent launch and waits for room: with room for ONE class at a time the result is bit-identical to the
    unconstrained run."""
    from smrt_amd.balance import smrt_balance
    monkeypatch.setenv("SMRT_HIP_FUSED", "1")
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1")
    X, y = _data(0)
    kw = dict(COMMON, **CALLS[0])
    outs = []
    for budget in (None, "40", "40"):
        if budget is None:
            monkeypatch.delenv("SMRT_HIP_PERSIST_BUDGET", raising=False)
        else:
            monkeypatch.setenv("SMRT_HIP_PERSIST_BUDGET", budget)         # 8 + 28 workgroups per class: one class at a time
        Xb, yb, enc = smrt_balance(X, y, **kw)
        assert [e.sess.last_epoch_launches for e in enc.values() if e is not None] == [2] * 4
        outs.append(Xb)
        for e in enc.values():
            if e is not None:
                e.clean_session()
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[1], outs[2])
