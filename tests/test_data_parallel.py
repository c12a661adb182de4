"""Data-parallel path: stripe plan, the host loop over 2 gloo ranks on CPU with an
oracle-backed session, and -- on the GPU -- the two C-ABI halves against the fused step and a 2-process fit
(both ranks on cuda:0, gloo) against the single-process fit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle.vae_oracle import OracleVAE, flatten_params, unflatten_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleDPSession(object):
    """Duck-types the VaeHandle methods DataParallelTrainer uses, on the CPU oracle (test double)."""

    def __init__(self, oracle):
        import torch
        self.o = oracle
        self.device = torch.device("cpu")
        self.n_params = flatten_params(oracle.params).size
        self.dp_count = self.n_params + 4

    def use_stream(self, stream=None):
        pass

    def dp_grad_step(self, X_local, off, rows, global_rows, row0, grads):
        import torch
        o = self.o
        grads.zero_()
        if rows == 0:
            return
        X = X_local[0][off:off + rows, :o.D].numpy()
        cache = o.forward(X, o.draw_eps(rows, row0))
        l2, o.l2 = o.l2, None                                # L2 is added once, after the reduction
        try:
            gW, gb = o.gradients(cache, inv_batch=1.0 / global_rows)
            loss = float(o.cost(cache)) * rows / global_rows
        finally:
            o.l2 = l2
        flat = np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(gW, gb)])
        grads[:self.n_params] = torch.from_numpy(flat.astype(np.float32))
        grads[self.n_params] = loss

    def dp_apply(self, grads, cost_view=None):
        o, dt, L = self.o, self.o.dt, self.o.L
        g = unflatten_params(grads[:self.n_params].numpy(), o.D, o.hidden, o.Z, dtype=np.float32)
        reg = list(range(L)) + list(range(L + 2, 2 * L + 3))
        cost = dt(grads[self.n_params].item())
        gW, gb = [w for w, _ in g], [b for _, b in g]
        if o.l2 is not None:
            cost = cost + o.l2 * sum(dt(0.5) * np.sum(o.params[i][0] ** 2) for i in reg)
            for i in reg:
                gW[i] = (gW[i] + o.l2 * o.params[i][0]).astype(dt)
        gW, gb = o.clip_grads(gW, gb)
        variables, gl = [], []
        for (W, b), a, c in zip(o.params, gW, gb):
            variables += [W, b]
            gl += [a, c]
        o.opt.apply(variables, gl)
        o.step += 1
        if cost_view is not None:
            cost_view[0] = float(cost)


def test_stripe_plan_covers_every_row_once():
    from smrt_amd.autoencode.data_parallel import local_row_index, plan_stripes
    for n, b, w in [(10, 4, 3), (1000, 128, 8), (5, 8, 2), (7, 2, 4), (1, 1, 3)]:
        plan = plan_stripes(n, b, w)
        assert [p[0] for p in plan] == list(range(0, n, b))                 # gen_batches starts
        assert sum(p[1] for p in plan) == n
        parts = [local_row_index(n, b, w, r) for r in range(w)]
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(n))
        for start, rows, bounds in plan:
            assert bounds[0] == start and bounds[-1] == start + rows and np.all(np.diff(bounds) >= 0)
            assert np.diff(bounds).max() - np.diff(bounds).min() <= 1         # balanced stripes
    with pytest.raises(ValueError):
        plan_stripes(0, 4, 2)


CPU_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from oracle.vae_oracle import OracleVAE, flatten_params
from smrt_amd.autoencode.data_parallel import DataParallelTrainer, local_row_index
from tests.test_data_parallel import OracleDPSession

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
d = np.load(os.path.join(out, "in.npz"))
X, batch, epochs = d["X"], int(d["batch"]), int(d["epochs"])
o = OracleVAE(X.shape[1], [9, 7], 3, activation="tanh", learning_function="adam", learning_rate=0.01, seed=5)
sess = OracleDPSession(o)
idx = local_row_index(len(X), batch, world, rank)
Xl = torch.from_numpy(np.ascontiguousarray(X[idx], dtype=np.float32))
tr = DataParallelTrainer(sess, (Xl, Xl.shape[1]), len(X), batch, rank, world, dist)
costs, secs = tr.run_epochs(epochs, early_stopping=True, min_change=1e-7)      # broadcast of the cost every epoch
np.savez(os.path.join(out, "out%d.npz" % rank), costs=costs, params=flatten_params(o.params), n_local=len(idx))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run_ranks(script, tmp_path, world=2, timeout=300):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port, str(tmp_path)]) for r in range(world)]
    for pr in procs:
        assert pr.wait(timeout=timeout) == 0


def test_dp_host_loop_two_gloo_ranks_equals_single_oracle(tmp_path):
    """The host loop (stripes, all-reduce, clip after the reduction, L2 once, last-batch cost) on 2 CPU ranks
    reproduces the single-process oracle fit; 37 rows / batch 8 leaves a ragged tail of 5."""
    rs = np.random.RandomState(3)
    X = rs.uniform(0, 1, size=(37, 11))
    np.savez(tmp_path / "in.npz", X=X, batch=8, epochs=3)
    ref = OracleVAE(11, [9, 7], 3, activation="tanh", learning_function="adam", learning_rate=0.01, seed=5)
    ref.fit(X, n_epochs=3, batch_size=8, early_stopping=True, min_change=1e-7)
    script = tmp_path / "worker.py"
    script.write_text(CPU_WORKER.format(root=ROOT))
    _run_ranks(script, tmp_path)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(2)]
    assert int(outs[0]["n_local"]) + int(outs[1]["n_local"]) == 37
    for d in outs:
        np.testing.assert_allclose(d["costs"], np.asarray(ref.epoch_costs_, np.float32), rtol=2e-5)
        np.testing.assert_allclose(d["params"], flatten_params(ref.params), rtol=1e-4, atol=2e-6)
    np.testing.assert_array_equal(outs[0]["params"], outs[1]["params"])       # replicas stay bit-identical


# ---------------------------------------------------------------------------------------- GPU
def _close(a, b, what, rtol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * max(np.abs(b).max(), 1e-30), err_msg=what)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["adam", "rms_prop"])
def test_grad_step_plus_apply_equals_oracle_step(opt):
    """smrt_vae_grad_step on two stripes + a host sum + smrt_vae_apply_grads == one oracle step on the whole batch."""
    import torch
    from smrt_amd import _hip
    rs = np.random.RandomState(1)
    D, hidden, Z, B = 20, [24, 12], 5, 50
    o = OracleVAE(D, hidden, Z, activation="sigmoid", learning_function=opt, learning_rate=0.02, seed=9)
    o.params = [((W + 0.05 * rs.normal(size=W.shape)).astype(np.float32), b) for W, b in o.params]
    hs = [_hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer=opt,
                         learning_rate=0.02, l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10,
                         dropout=1.0, seed=9, max_batch=B, device=0) for _ in range(2)]
    for h in hs:
        h.set_params(flatten_params(o.params))
    dev = hs[0].device
    for step in range(3):
        X = rs.uniform(0, 1, size=(B, D))
        bufs = []
        for r, (lo, hi) in enumerate([(0, 21), (21, B)]):
            g = torch.zeros(hs[r].dp_count, dtype=torch.float32, device=dev)
            hs[r].grad_step(X[lo:hi], B, lo, g)                      # Philox epsilon indexed by row in the batch
            bufs.append(g)
        total = bufs[0] + bufs[1]
        torch.cuda.synchronize(dev)
        costs = [h.apply_grads(total) for h in hs]
        c_o = o.train_step(X)                                        # draws the same Philox epsilon (seed 9, step)
        assert costs[0] == costs[1]
        assert np.isclose(costs[0], c_o, rtol=1e-4), (step, costs[0], c_o)
    for h in hs:
        _close(h.get_params(), flatten_params(o.params), "params after 3 DP steps")
        h.close()


GPU_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from smrt_amd.autoencode import VariationalAutoEncoder

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(0)
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
d = np.load(os.path.join(out, "in.npz"))
X = d["X"]
est = VariationalAutoEncoder([48, 32], 6, activation_function="relu", learning_function="adam", learning_rate=0.01,
                             n_epochs=3, batch_size=256, random_state=7, early_stopping=True, min_change=1e-7)
est.data_parallel = True
est.fit(X)
np.savez(os.path.join(out, "out%d.npz" % rank), costs=np.asarray(est.epoch_costs_),
         params=est.sess.get_params(), gen=est.generate(5))
est.clean_session()
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_dp_fit_two_ranks_on_one_gpu_equals_single_fit(tmp_path):
    """VariationalAutoEncoder.fit with data_parallel on 2 ranks (gloo, both on cuda:0) == the single-process fit."""
    from smrt_amd.autoencode import VariationalAutoEncoder
    rs = np.random.RandomState(11)
    X = rs.uniform(0, 1, size=(1100, 40))                             # batch 256 -> ragged tail of 76
    np.savez(tmp_path / "in.npz", X=X)
    single = VariationalAutoEncoder([48, 32], 6, activation_function="relu", learning_function="adam",
                                    learning_rate=0.01, n_epochs=3, batch_size=256, random_state=7, early_stopping=True,
                                    min_change=1e-7).fit(X)
    ref_params, ref_costs, ref_gen = single.sess.get_params(), np.asarray(single.epoch_costs_), single.generate(5)
    single.clean_session()
    script = tmp_path / "worker.py"
    script.write_text(GPU_WORKER.format(root=ROOT))
    _run_ranks(script, tmp_path, timeout=600)
    outs = [np.load(tmp_path / ("out%d.npz" % r)) for r in range(2)]
    for d in outs:
        np.testing.assert_allclose(d["costs"], ref_costs, rtol=1e-4)
        _close(d["params"], ref_params, "DP params vs single-GPU params", rtol=2e-4)
        _close(d["gen"], ref_gen, "generate after DP fit", rtol=2e-4)
    np.testing.assert_array_equal(outs[0]["params"], outs[1]["params"])
