"""CPU-only checks: the C-ABI library loads and exports every symbol include/smrt_hip.h declares, the
product path fails loudly without a GPU (no fallback), and the N>1 per-class sharding of smrt_balance
(world_size 2, gloo) returns exactly the single-process result."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "smrt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(smrt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from smrt_amd import _hip
    assert os.path.exists(_hip.lib_path()), "run __graft_entry__.build() first"
    import torch  # noqa: F401  (same load order as the product: torch's HIP runtime first)
    lib = ctypes.CDLL(_hip.lib_path())
    declared = _header_functions()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_hip.EXPORTS) == declared            # the ctypes binding covers the whole header
    lib.smrt_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.smrt_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from smrt_amd import _hip
    from smrt_amd.autoencode import VariationalAutoEncoder
    with pytest.raises(_hip.HipLibraryError):
        VariationalAutoEncoder(4, 2, n_epochs=1).fit(np.random.RandomState(0).uniform(size=(8, 6)))


def test_plan_class_jobs_longest_first():
    from smrt_amd.balance.smrt import plan_class_jobs
    assert plan_class_jobs([5, 5, 5], 1) == [0, 0, 0]
    owners = plan_class_jobs([10, 1, 1, 9, 2], 2)
    load = [sum(c for c, o in zip([10, 1, 1, 9, 2], owners) if o == r) for r in range(2)]
    assert sorted(load) == [11, 12] and owners[0] != owners[3]
    assert plan_class_jobs([3, 3, 3, 3], 4) == [0, 1, 2, 3]
    assert plan_class_jobs([], 8) == []


WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
from smrt_amd.balance import smrt_balance
from tests.test_host_goldens import FakeVAE

rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
d = np.load(os.path.join(out, "in.npz"))
X, y = d["X"], d["y"]
res = dict()
for name, kw in (("gen", dict()), ("samples", dict(gen_from_samples=True))):
    prefit = dict((lab, FakeVAE(i + 1, X.shape[1])) for i, lab in enumerate([1, 2, 3]))
    Xb, yb, enc = smrt_balance(X, y, 5, 2, balance_ratio=0.6, random_state=13, shuffle=True, prefit_estimators=prefit,
                               return_estimators=True, **kw)
    res[name + "_X"], res[name + "_y"] = Xb, yb
    res[name + "_owned"] = np.array([enc[k] is not None for k in (1, 2, 3)])
np.savez(os.path.join(out, "out%d.npz" % rank), **res)
dist.barrier()
dist.destroy_process_group()
"""


def test_smrt_balance_shards_classes_across_ranks(tmp_path):
    from smrt_amd.balance import smrt_balance
    from tests.test_host_goldens import FakeVAE
    rs = np.random.RandomState(0)
    y = np.array([0] * 50 + [1] * 12 + [2] * 7 + [3] * 20)
    X = rs.uniform(size=(len(y), 3))
    p = rs.permutation(len(y))
    X, y = X[p], y[p]
    np.savez(tmp_path / "in.npz", X=X, y=y)
    single = {}
    for name, kw in (("gen", dict()), ("samples", dict(gen_from_samples=True))):
        prefit = {lab: FakeVAE(i + 1, 3) for i, lab in enumerate([1, 2, 3])}
        single[name] = smrt_balance(X, y, 5, 2, balance_ratio=0.6, random_state=13, shuffle=True,
                                    prefit_estimators=prefit, **kw)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(tmp_path)]) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=180) == 0
    owned = []
    for r in range(2):
        d = np.load(tmp_path / ("out%d.npz" % r))
        for name in ("gen", "samples"):
            np.testing.assert_array_equal(d[name + "_X"], single[name][0])
            np.testing.assert_array_equal(d[name + "_y"], single[name][1])
        owned.append(d["gen_owned"])
    # every class job ran on exactly one rank, and both ranks got work
    assert np.array_equal(owned[0] ^ owned[1], np.ones(3, bool)) and owned[0].any() and owned[1].any()


def test_assemble_rows_matches_numpy():
    """smrt_assemble_rows (host side of the library): fp64([X ; synth][order]) for fp32 / fp64 / other inputs, identity
    order, no synthetic rows, many threads vs one, and rejected out-of-range row numbers."""
    from smrt_amd import _hip
    rs = np.random.RandomState(4)
    S = rs.uniform(size=(37, 5)).astype(np.float32)
    for X in (rs.uniform(size=(101, 5)), rs.uniform(size=(101, 5)).astype(np.float32), rs.randint(0, 9, size=(101, 5))):
        ref = np.vstack([np.asarray(X, np.float64), S.astype(np.float64)])
        order = rs.permutation(138)
        for nt in (1, 3):
            got = _hip.assemble_rows(X, S, order, nt)
            assert got.dtype == np.float64 and np.array_equal(got, ref[order])
        assert np.array_equal(_hip.assemble_rows(X, S, None), ref)
        assert np.array_equal(_hip.assemble_rows(X, S[:0], None), ref[:101])
        assert np.array_equal(_hip.assemble_rows(X, S, order[:10]), ref[order[:10]])
    big = rs.uniform(size=(20000, 3)).astype(np.float32)                 # enough rows for several threads
    order = rs.permutation(20037)
    assert np.array_equal(_hip.assemble_rows(big, S[:, :3], order, 8),
                          np.vstack([big.astype(np.float64), S[:, :3].astype(np.float64)])[order])
    with pytest.raises(_hip.HipLibraryError):
        _hip.assemble_rows(big, S[:, :3], np.array([0, 20037]))
    with pytest.raises(ValueError):
        _hip.assemble_rows(big, S, None)
    # the two halves (rows from X first, synthetic rows later) add up to the one-pass result, and touch nothing else
    for order_ in (order, None):
        sa = _hip.SplitAssembly(big, 37, order_, n_threads=4)
        sa.out[:] = -7.0
        sa.originals()
        ref = np.vstack([big.astype(np.float64), S[:, :3].astype(np.float64)])
        ref = ref if order_ is None else ref[order_]
        from_x = (np.arange(20037) if order_ is None else order_) < 20000
        assert np.array_equal(sa.out[from_x], ref[from_x]) and np.all(sa.out[~from_x] == -7.0)
        assert np.array_equal(sa.synthetic(S[:, :3]), ref)
    with pytest.raises(ValueError):
        _hip.SplitAssembly(big, 37, None).synthetic(S[:5, :3])
    # the synthetic half in pieces on the background thread (smrt_balance: a class's rows go in while the next class's are
    # downloaded): each piece touches only its rows, together they are the one-pass result; many rows, several threads
    S3 = np.ascontiguousarray(rs.uniform(size=(30011, 3)).astype(np.float32))
    order_big = rs.permutation(50011)
    for order_ in (order_big, None):
        sa = _hip.SplitAssembly(big, 30011, order_, n_threads=5)
        sa.out[:] = -7.0
        sa.originals()
        ref = np.vstack([big.astype(np.float64), S3.astype(np.float64)])
        ref = ref if order_ is None else ref[order_]
        src = np.arange(50011) if order_ is None else order_
        sa.synthetic_range_async(S3, 100, 20000)
        assert sa.wait() is sa.out
        mine = (src >= 20100) & (src < 40000)
        assert np.array_equal(sa.out[mine], ref[mine]) and np.all(sa.out[(src >= 20000) & ~mine] == -7.0)
        for lo, hi in ((0, 100), (20000, 20000), (20000, 30011)):
            sa.synthetic_range_async(S3, lo, hi)
        assert np.array_equal(sa.wait(), ref)
        with pytest.raises(_hip.HipLibraryError):
            sa.synthetic_range_async(S3, 5, 30012)
            sa.wait()
        assert np.array_equal(sa.synthetic(S3), ref)


def test_all_finite_sweep_and_validation_error_text():
    """smrt_all_finite (the multi-threaded stand-in for check_array's finiteness pass): NaN / +-inf anywhere, fp32 and
    fp64, largest finite values, and validation still raising sklearn's own message."""
    from sklearn.utils.validation import check_array
    from smrt_amd import _hip
    from smrt_amd.balance.base import validate_X_y_ratio_classes
    rs = np.random.RandomState(0)
    for dtype, big in ((np.float32, 3.4e38), (np.float64, 1.7e308)):
        X = rs.uniform(-1, 1, size=(70001, 16)).astype(dtype)
        X[0, 0], X[-1, -1] = big, -big
        for nt in (1, 5):
            assert _hip.all_finite(X, nt)
        for bad in (np.nan, np.inf, -np.inf):
            for pos in ((0, 0), (35000, 7), (70000, 15)):
                Y = X.copy()
                Y[pos] = bad
                assert not _hip.all_finite(Y, 4)
    X = rs.uniform(size=(70000, 16)).astype(np.float32)               # >= 2^20 elements: the fast path of validation
    y = (np.arange(70000) % 7 == 0).astype(int)
    assert validate_X_y_ratio_classes(X, y, 0.5, copy=False, keep_float32=True)[0] is X
    X[123, 4] = np.nan
    with pytest.raises(ValueError) as ours:
        validate_X_y_ratio_classes(X, y, 0.5)
    with pytest.raises(ValueError) as theirs:
        check_array(X)
    assert str(ours.value) == str(theirs.value)


def test_reference_import_name_is_an_alias():
    """`from smrt.balance import smrt_balance` (smrt/balance/tests/test_smrt.py:9) works unchanged: `smrt` is the same
    module objects as `smrt_amd`, not a second implementation."""
    import smrt
    import smrt_amd.autoencode
    import smrt_amd.balance
    from smrt.autoencode import AutoEncoder, VariationalAutoEncoder
    from smrt.autoencode.autoencoder import VariationalAutoEncoder as V2
    from smrt.balance import over_sample_balance, smote_balance, smrt_balance, under_sample_balance
    from smrt.balance.smrt import smrt_balance as sb2
    from smrt.utils import DEFAULT_SEED, get_random_state
    assert smrt_balance is smrt_amd.balance.smrt_balance is sb2
    assert VariationalAutoEncoder is smrt_amd.autoencode.VariationalAutoEncoder is V2
    assert AutoEncoder is smrt_amd.autoencode.AutoEncoder
    assert smrt.balance is smrt_amd.balance and DEFAULT_SEED == 42
    assert callable(over_sample_balance) and callable(smote_balance) and callable(under_sample_balance)
    assert get_random_state(7).seed_value == 7
