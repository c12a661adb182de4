"""HIP path vs the CPU oracle, through the C ABI (smrt_amd._hip -> libsmrt_hip.so).

Tolerance: north_star asks for 1e-4 relative in fp32 on per-step loss, gradients and generated
samples with injected initial weights and epsilon.  Gradients are compared with
``rtol=1e-4, atol=1e-4 * max|g|`` per tensor (relative to the tensor's scale: single entries can be
the difference of large cancelling terms).
"""
import numpy as np
import pytest

from oracle.vae_oracle import (ACTIVATIONS, OracleVAE, flatten_params, philox_normal, unflatten_params,
                               STREAM_EPS, STREAM_GENERATE)

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(autouse=True, params=["fused", "per_layer"])
def kernel_path(request, monkeypatch):
    """Every case runs twice: with the row-tile fused forward/backward kernel (taken when the net fits in
    LDS) and with the per-layer GEMM kernels (always taken by large nets)."""
    monkeypatch.setenv("SMRT_HIP_FUSED", "1" if request.param == "fused" else "0")
    return request.param


def _handle(D, hidden, Z, act="sigmoid", opt="rms_prop", lr=0.05, l2=1e-4, clip=True, drop=1.0, seed=42,
            max_batch=256, eps=1e-10, gmin=-5.0, gmax=5.0):
    from smrt_amd import _hip
    return _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation=act, optimizer=opt, learning_rate=lr,
                          l2_penalty=l2, clip=clip, gclip_min=gmin, gclip_max=gmax, eps=eps, dropout=drop, seed=seed,
                          max_batch=max_batch, device=0)


def _oracle(D, hidden, Z, act="sigmoid", opt="rms_prop", lr=0.05, l2=1e-4, clip=True, drop=1.0, seed=42,
            eps=1e-10, gmin=-5.0, gmax=5.0, rs=None, dtype=np.float32):
    o = OracleVAE(D, hidden, Z, activation=act, learning_function=opt, learning_rate=lr, l2_penalty=l2, clip=clip,
                  gclip_min=gmin, gclip_max=gmax, eps=eps, dropout=drop, seed=seed, dtype=dtype)
    if rs is not None:      # decorrelate the identical heads, move biases off zero
        o.params = [((W + 0.05 * rs.normal(size=W.shape)).astype(dtype), (b + 0.05 * rs.normal(size=b.shape)).astype(dtype))
                    for W, b in o.params]
    return o


def _close(a, b, what, rtol=RTOL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * scale, err_msg=what)


def _flat_grads(gW, gb):
    return np.concatenate([np.concatenate([w.ravel(), b.ravel()]) for w, b in zip(gW, gb)])


def _per_layer(flat, D, hidden, Z):
    return unflatten_params(flat, D, hidden, Z)


@pytest.mark.parametrize("act", ACTIVATIONS)
@pytest.mark.parametrize("opt", ["rms_prop", "adam", "sgd"])
def test_three_steps_tiny(act, opt):
    """Tiny ragged net (D=12, h=7, Z=3, B=5 then B=3): cost, every gradient, weights after 3 steps."""
    rs = np.random.RandomState(0)
    D, hidden, Z = 12, [7], 3
    o = _oracle(D, hidden, Z, act=act, opt=opt, rs=rs)
    h = _handle(D, hidden, Z, act=act, opt=opt, max_batch=8)
    h.set_params(flatten_params(o.params))
    _close(h.get_params(), flatten_params(o.params), "round trip", rtol=0)
    for step, B in enumerate([5, 3, 5]):
        X = rs.uniform(0, 1, size=(B, D))
        eps = rs.normal(size=(B, Z))
        c_o, (gW, gb) = o.train_step(X, eps, return_grads=True)
        c_h, g_h = h.train_step(X, eps, want_grads=True)
        assert np.isclose(c_h, c_o, rtol=RTOL), (step, c_h, c_o)
        for (hw, hb), w, b in zip(_per_layer(g_h, D, hidden, Z), gW, gb):
            _close(hw, w, "dW step %d" % step)
            _close(hb, b, "db step %d" % step)
    for (hw, hb), (w, b) in zip(_per_layer(h.get_params(), D, hidden, Z), o.params):
        _close(hw, w, "W after 3 steps")
        _close(hb, b, "b after 3 steps")
    assert h.step_count == 3
    h.close()


@pytest.mark.parametrize("opt", ["adagrad", "adadelta", "proximal-sgd", "proximal-adagrad"])
def test_remaining_optimizers(opt):
    rs = np.random.RandomState(1)
    D, hidden, Z = 10, [6, 5], 2
    o = _oracle(D, hidden, Z, act="tanh", opt=opt, lr=0.1, rs=rs)
    h = _handle(D, hidden, Z, act="tanh", opt=opt, lr=0.1, max_batch=8)
    h.set_params(flatten_params(o.params))
    for _ in range(3):
        X, eps = rs.uniform(0, 1, size=(6, D)), rs.normal(size=(6, Z))
        c_o = o.train_step(X, eps)
        c_h = h.train_step(X, eps)
        assert np.isclose(c_h, c_o, rtol=RTOL)
    for (hw, hb), (w, b) in zip(_per_layer(h.get_params(), D, hidden, Z), o.params):
        _close(hw, w, "W")
        _close(hb, b, "b")
    slots = o.opt.state_arrays()
    first = {"adagrad": "acc", "proximal-adagrad": "acc", "adadelta": "acc"}.get(opt)
    if first:
        got = _per_layer(h.get_opt_state(0), D, hidden, Z)
        for i, (hw, hb) in enumerate(got):
            _close(hw, slots[first][2 * i], "slot0 W")
            _close(hb, slots[first][2 * i + 1], "slot0 b")
    h.close()


@pytest.mark.parametrize("clip,l2", [(False, None), (True, None), (False, 1e-2)])
def test_clip_and_l2_switches(clip, l2):
    rs = np.random.RandomState(2)
    D, hidden, Z = 16, [9], 4
    # tight clip bounds so the clip is active
    o = _oracle(D, hidden, Z, act="relu", opt="sgd", l2=l2, clip=clip, gmin=-1e-3, gmax=2e-3, rs=rs)
    h = _handle(D, hidden, Z, act="relu", opt="sgd", l2=l2, clip=clip, gmin=-1e-3, gmax=2e-3, max_batch=16)
    h.set_params(flatten_params(o.params))
    X, eps = rs.uniform(0, 1, size=(11, D)), rs.normal(size=(11, Z))
    c_o, (gW, gb) = o.train_step(X, eps, return_grads=True)
    c_h, g_h = h.train_step(X, eps, want_grads=True)
    assert np.isclose(c_h, c_o, rtol=RTOL)
    _close(g_h, _flat_grads(gW, gb), "pre-clip gradient")
    _close(h.get_params(), flatten_params(o.params), "params after clipped step")
    h.close()


def test_dropout_injected_masks():
    rs = np.random.RandomState(3)
    D, hidden, Z = 12, [8], 3
    o = _oracle(D, hidden, Z, act="sigmoid", opt="rms_prop", drop=0.5, rs=rs)
    h = _handle(D, hidden, Z, act="sigmoid", opt="rms_prop", drop=0.5, max_batch=8)
    h.set_params(flatten_params(o.params))
    for _ in range(2):
        masks = [np.floor(0.5 + rs.uniform(size=W.shape)).astype(np.float32) for W, _ in o.params]
        h.set_dropout_masks(flatten_params([(m, np.ones_like(b)) for m, (_, b) in zip(masks, o.params)]))
        X, eps = rs.uniform(0, 1, size=(7, D)), rs.normal(size=(7, Z))
        c_o, (gW, gb) = o.train_step(X, eps, masks=masks, return_grads=True)
        c_h, g_h = h.train_step(X, eps, want_grads=True)
        assert np.isclose(c_h, c_o, rtol=RTOL)
        _close(g_h, _flat_grads(gW, gb), "gradient through dropout")
    _close(h.get_params(), flatten_params(o.params), "params")
    h.close()


def test_dropout_philox_stream_matches_oracle():
    rs = np.random.RandomState(4)
    D, hidden, Z = 12, [8], 3
    o = _oracle(D, hidden, Z, act="tanh", opt="sgd", drop=0.4, seed=99, rs=rs)
    h = _handle(D, hidden, Z, act="tanh", opt="sgd", drop=0.4, seed=99, max_batch=8)
    h.set_params(flatten_params(o.params))
    for _ in range(2):
        X, eps = rs.uniform(0, 1, size=(6, D)), rs.normal(size=(6, Z))
        c_o = o.train_step(X, eps)            # oracle draws its masks from the same Philox stream
        c_h = h.train_step(X, eps)
        assert np.isclose(c_h, c_o, rtol=RTOL)
    _close(h.get_params(), flatten_params(o.params), "params")
    h.close()


def test_philox_epsilon_stream():
    h = _handle(8, [4], 5, seed=1234567890123, max_batch=8)
    for step, row0, rows in [(0, 0, 7), (3, 11, 4), (70000, 123456789, 3)]:
        got = h.sample_eps(step, row0, rows)
        want = philox_normal(1234567890123, STREAM_EPS, step, row0 * 5, rows * 5).reshape(rows, 5)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    big = h.sample_eps(1, 0, 8)
    h.close()
    h2 = _handle(8, [4], 5, seed=7, max_batch=8)
    n = np.concatenate([h2.sample_eps(s, 0, 4000).ravel() for s in range(5)])
    assert abs(n.mean()) < 0.02 and abs(n.std() - 1) < 0.02
    h2.close()


def test_train_with_kernel_epsilon_matches_oracle():
    """No injection: the kernels draw epsilon from Philox; the oracle mirrors the stream."""
    rs = np.random.RandomState(5)
    D, hidden, Z = 20, [11], 4
    o = _oracle(D, hidden, Z, act="elu", opt="adam", lr=0.01, seed=2024, rs=rs)
    h = _handle(D, hidden, Z, act="elu", opt="adam", lr=0.01, seed=2024, max_batch=16)
    h.set_params(flatten_params(o.params))
    for B in (16, 9, 16):
        X = rs.uniform(0, 1, size=(B, D))
        assert np.isclose(h.train_step(X), o.train_step(X), rtol=RTOL)
    _close(h.get_params(), flatten_params(o.params), "params")
    h.close()


def test_n1_single_step_and_inference():
    """MNIST-shaped net (784-450-(10,10)-450-784), B=256 and the ragged B=244 of config 1."""
    rs = np.random.RandomState(6)
    D, hidden, Z = 784, [450], 10
    o = _oracle(D, hidden, Z, rs=rs)
    h = _handle(D, hidden, Z, max_batch=256)
    h.set_params(flatten_params(o.params))
    for B in (256, 244):
        X = (rs.uniform(0, 1, size=(B, D)) * (rs.uniform(size=(B, D)) > 0.8)).astype(np.float64)
        eps = rs.normal(size=(B, Z))
        c_o, (gW, gb) = o.train_step(X, eps, return_grads=True)
        c_h, g_h = h.train_step(X, eps, want_grads=True)
        assert np.isclose(c_h, c_o, rtol=RTOL), (c_h, c_o)
        for (hw, hb), w, b in zip(_per_layer(g_h, D, hidden, Z), gW, gb):
            _close(hw, w, "dW")
            _close(hb, b, "db")
    _close(h.get_params(), flatten_params(o.params), "params")
    Xe = rs.uniform(0, 1, size=(300, D))
    mu, ls = h.encode(Xe)
    mu_o, ls_o = o.encode(Xe)
    _close(mu, mu_o, "z_mean")
    _close(ls, ls_o, "z_log_sigma")
    Zs = rs.normal(size=(333, Z))
    _close(h.decode(Zs), o.decode(Zs), "decode")
    h.close()


def test_two_hidden_layers_odd_widths():
    rs = np.random.RandomState(7)
    D, hidden, Z = 67, [45, 130], 9
    o = _oracle(D, hidden, Z, act="tanh", opt="rms_prop", rs=rs)
    h = _handle(D, hidden, Z, act="tanh", opt="rms_prop", max_batch=200)
    h.set_params(flatten_params(o.params))
    for B in (200, 131, 1):
        X, eps = rs.uniform(0, 1, size=(B, D)), rs.normal(size=(B, Z))
        c_o, (gW, gb) = o.train_step(X, eps, return_grads=True)
        c_h, g_h = h.train_step(X, eps, want_grads=True)
        assert np.isclose(c_h, c_o, rtol=RTOL)
        _close(g_h, _flat_grads(gW, gb), "grads B=%d" % B)
    _close(h.get_params(), flatten_params(o.params), "params")
    Zs = rs.normal(size=(5, Z))
    _close(h.decode(Zs), o.decode(Zs), "decode")
    assert h.decode(np.zeros((0, Z))).shape == (0, D)              # empty inputs
    mu0, ls0 = h.encode(np.zeros((0, D)))
    assert mu0.shape == (0, Z) and ls0.shape == (0, Z)
    h.close()


@pytest.mark.parametrize("D,hidden,Z,act,opt", [
    (100, [72], 5, "sigmoid", "rms_prop"),      # k extents 112 (4 + 2 + 1 blocks of 16), 80 (4 + 1), 16 (1)
    (24, [176], 8, "tanh", "adam"),             # k extents 32 (2), 176 (8 + 2 + 1), 16 (1)
])
def test_fused_small_net_every_k_chunk(D, hidden, Z, act, opt, kernel_path):
    """The LDS-resident two-kernel step walks k in compile-time chunks of 8 / 4 / 2 / 1 blocks of 16: nets whose padded
    widths need every combination, at a full, a ragged and a one-row batch, with injected epsilon, vs the oracle step."""
    rs = np.random.RandomState(19)
    o = _oracle(D, hidden, Z, act=act, opt=opt, lr=0.01, rs=rs)
    h = _handle(D, hidden, Z, act=act, opt=opt, lr=0.01, max_batch=48)
    h.set_params(flatten_params(o.params))
    for B in (48, 37, 1):
        X, eps = rs.uniform(0, 1, size=(B, D)), rs.normal(size=(B, Z))
        c_o, (gW, gb) = o.train_step(X, eps, return_grads=True)
        c_h, g_h = h.train_step(X, eps, want_grads=True)
        assert np.isclose(c_h, c_o, rtol=RTOL)
        _close(g_h, _flat_grads(gW, gb), "grads B=%d" % B)
        if kernel_path == "fused":
            assert h.last_epoch_launches <= 4, "expected the two-kernel fused step (+ image build, + finalize)"
    _close(h.get_params(), flatten_params(o.params), "params")
    h.close()


def test_bce_saturation_matches_ieee():
    """eps=1e-10: float32(1-eps) == 1.0f, so p == 1.0f is not clipped (inf / NaN cost exactly as IEEE says)."""
    D, hidden, Z = 4, [3], 2
    o = _oracle(D, hidden, Z, act="identity", opt="sgd", l2=None)
    big = [(np.zeros_like(W), np.zeros_like(b)) for W, b in o.params]
    big[-1] = (big[-1][0], np.array([40.0, -200.0, 40.0, 0.0], np.float32))    # p = 1.0f, ~0, 1.0f, .5
    o.params = big
    h = _handle(D, hidden, Z, act="identity", opt="sgd", l2=None, max_batch=4)
    h.set_params(flatten_params(big))
    X = np.array([[1.0, 0.0, 0.0, 1.0]])
    eps = np.zeros((1, Z))
    with np.errstate(all="ignore"):
        c_o = o.train_step(X.copy(), eps)
    c_h = h.train_step(X, eps)
    assert (np.isnan(c_o) and np.isnan(c_h)) or (np.isinf(c_o) and np.isinf(c_h) and np.sign(c_o) == np.sign(c_h)), (c_o, c_h)
    # ... and what the step leaves behind: inf * 0 in the output layer is a NaN gradient, which tf.clip_by_value hands on
    # (std::max / std::min keep a NaN first argument; fmaxf / fminf would turn it into +-5): the same parameters are NaN on
    # both sides, the others agree, and the next step's cost is NaN
    p_h, p_o = h.get_params(), flatten_params(o.params)
    assert np.isnan(p_o).any() and np.array_equal(np.isnan(p_h), np.isnan(p_o)), (p_h, p_o)
    np.testing.assert_allclose(p_h[~np.isnan(p_o)], p_o[~np.isnan(p_o)], rtol=RTOL, atol=1e-6)
    with np.errstate(all="ignore"):
        assert np.isnan(o.train_step(X.copy(), eps)) and np.isnan(h.train_step(X, eps))
    h.close()
    # lower clip: p underflows to < eps -> log(eps), zero gradient through the clip
    o2 = _oracle(D, hidden, Z, act="identity", opt="sgd", l2=None)
    p2 = [(np.zeros_like(W), np.zeros_like(b)) for W, b in o2.params]
    p2[-1] = (p2[-1][0], np.array([-50.0, -50.0, 0.0, 0.0], np.float32))
    o2.params = p2
    h2 = _handle(D, hidden, Z, act="identity", opt="sgd", l2=None, max_batch=4)
    h2.set_params(flatten_params(p2))
    X2 = np.array([[1.0, 0.0, 0.5, 0.5]])
    c_o2, (gW, gb) = o2.train_step(X2.copy(), eps, return_grads=True)
    c_h2, g_h2 = h2.train_step(X2, eps, want_grads=True)
    assert np.isfinite(c_o2) and np.isclose(c_h2, c_o2, rtol=RTOL)
    _close(g_h2, _flat_grads(gW, gb), "grads at the lower clip")
    h2.close()


def test_train_epochs_schedule_matches_oracle_fit():
    """Whole _train loop on device (graph-captured epochs) vs the oracle's per-step loop: contiguous batches,
    ragged tail, last-batch cost per epoch."""
    rs = np.random.RandomState(8)
    D, hidden, Z, n, B = 24, [16], 4, 150, 64          # batches 64, 64, 22
    o = _oracle(D, hidden, Z, act="sigmoid", opt="rms_prop", seed=11, rs=rs)
    h = _handle(D, hidden, Z, act="sigmoid", opt="rms_prop", seed=11, max_batch=B)
    h.set_params(flatten_params(o.params))
    X = rs.uniform(0, 1, size=(n, D))
    o.fit(X, n_epochs=4, batch_size=B)
    costs, ms = h.train_epochs(X, B, 4)
    assert len(costs) == 4 and np.all(ms > 0)
    np.testing.assert_allclose(costs, np.array(o.epoch_costs_, np.float32), rtol=RTOL)
    _close(h.get_params(), flatten_params(o.params), "params after 4 epochs")
    assert h.step_count == 12
    h.close()


def test_early_stopping_rule():
    rs = np.random.RandomState(9)
    D, hidden, Z, n, B = 16, [8], 2, 64, 32
    o = _oracle(D, hidden, Z, opt="sgd", lr=1e-4, seed=5, rs=rs)
    h = _handle(D, hidden, Z, opt="sgd", lr=1e-4, seed=5, max_batch=B)
    h.set_params(flatten_params(o.params))
    X = rs.uniform(0, 1, size=(n, D))
    o.fit(X, n_epochs=30, batch_size=B, early_stopping=True, min_change=0.05)
    costs, _ = h.train_epochs(X, B, 30, early_stopping=True, min_change=0.05)
    assert len(costs) == len(o.epoch_costs_) < 30
    np.testing.assert_allclose(costs, np.array(o.epoch_costs_, np.float32), rtol=1e-4)
    h.close()


def test_fast_generate_stream():
    rs = np.random.RandomState(10)
    D, hidden, Z = 32, [20], 6
    o = _oracle(D, hidden, Z, seed=77, rs=rs)
    h = _handle(D, hidden, Z, seed=77, max_batch=8)
    h.set_params(flatten_params(o.params))
    got = h.generate(50, philox_offset=12, loc=0.5, scale=2.0)
    z = 0.5 + 2.0 * philox_normal(77, STREAM_GENERATE, 0, 12, 50 * Z).reshape(50, Z)
    _close(got, o.decode(z), "fast generate")
    h.close()


@pytest.mark.parametrize("Z", [4, 12, 24, 32])
def test_generate_draws_z_inside_the_launch_for_every_quad_count(Z):
    """Z % 4 == 0 and Z <= 32 with a block-aligned stream offset: each lane of fused_infer_kernel draws its own B fragments
    (one Philox block per lane and k-block; Z = 24 / 32 take a second k-block, Z = 4 / 12 leave quads of the first one
    empty).  Rows against the oracle's decode of the same Philox elements, at a row count with a ragged last 16-row block,
    and against the same rows drawn by the separate Philox launch (an unaligned offset one element further on)."""
    rs = np.random.RandomState(Z)
    D, hidden = 48, [96]
    o = _oracle(D, hidden, Z, seed=31, rs=rs)
    h = _handle(D, hidden, Z, seed=31, max_batch=16)
    h.set_params(flatten_params(o.params))
    n, off = 1000 + 7, 4 * 123
    got = h.generate(n, philox_offset=off, loc=-0.25, scale=1.25)
    z = -0.25 + 1.25 * philox_normal(31, STREAM_GENERATE, 0, off, n * Z).reshape(n, Z)
    _close(got, o.decode(z), "in-kernel draw, Z = %d" % Z)
    shifted = h.generate(n, philox_offset=off + 1, loc=-0.25, scale=1.25)          # separate draw (offset not block-aligned)
    z1 = -0.25 + 1.25 * philox_normal(31, STREAM_GENERATE, 0, off + 1, n * Z).reshape(n, Z)
    _close(shifted, o.decode(z1), "separate draw, Z = %d" % Z)
    h.close()


@pytest.mark.parametrize("D,hidden,Z,act,drop", [
    (64, [128], 16, "sigmoid", 1.0),        # the bench net N2: whole 16-byte row stores
    (67, [45, 130], 9, "tanh", 1.0),        # odd widths: element-wise stores, two encoder / three decoder layers
    (30, [24], 10, "elu", 0.5),             # dropout: the image twin is rebuilt from W (inference runs with keep = 1)
])
def test_fused_inference_launch_matches_oracle(D, hidden, Z, act, drop, kernel_path, monkeypatch):
    """encode / decode / generate of a net on the fused path run as ONE persistent launch (fused_infer_kernel: a wave
    carries 16 rows through every layer in LDS).  Checked against the oracle at row counts that leave a ragged last
    16-row block and give some waves several blocks, after a training step (image kept current by the optimizer
    epilogue), and against the per-layer GEMM chain on the same handle."""
    rs = np.random.RandomState(23)
    o = _oracle(D, hidden, Z, act=act, drop=drop, rs=rs)
    h = _handle(D, hidden, Z, act=act, drop=drop, max_batch=64)
    h.set_params(flatten_params(o.params))
    if drop == 1.0:
        X, eps = rs.uniform(0, 1, size=(64, D)), rs.normal(size=(64, Z))
        o.train_step(X, eps); h.train_step(X, eps)
    for n in (1, 16, 37, 5003):
        Xe, Zs = rs.uniform(0, 1, size=(n, D)), rs.normal(size=(n, Z))
        mu, ls = h.encode(Xe)
        mu_o, ls_o = o.encode(Xe)
        _close(mu, mu_o, "mu n=%d" % n); _close(ls, ls_o, "log_sigma n=%d" % n)
        dec = h.decode(Zs)
        _close(dec, o.decode(Zs), "decode n=%d" % n)
        if kernel_path == "fused":
            monkeypatch.setenv("SMRT_HIP_FUSED_INFER", "0")        # same handle, per-layer GEMM chain
            mu2, ls2 = h.encode(Xe)
            dec2 = h.decode(Zs)
            monkeypatch.delenv("SMRT_HIP_FUSED_INFER")
            _close(mu, mu2, "mu vs per-layer", rtol=1e-5); _close(ls, ls2, "log_sigma vs per-layer", rtol=1e-5)
            _close(dec, dec2, "decode vs per-layer", rtol=1e-5)
    got = h.generate(4099, philox_offset=3, loc=0.25, scale=1.5)
    z = 0.25 + 1.5 * philox_normal(42, STREAM_GENERATE, 0, 3, 4099 * Z).reshape(4099, Z)
    _close(got, o.decode(z), "generate")
    h.close()


def test_generate_across_z_chunks_is_position_independent():
    """smrt_vae_generate draws z in chunks (1 Mi rows on the fused path, 32 768 on the per-layer path); row r uses the
    Philox elements philox_offset + r * Z ..., whatever chunk or 16-row block it falls in.  Full-size property: the rows
    either side of every chunk boundary of one big call equal small calls that start at those rows."""
    import torch
    D, hidden, Z = 64, [128], 16
    n = (1 << 20) + 37
    h = _handle(D, hidden, Z, seed=5, max_batch=128)
    big = h.generate(n, philox_offset=11, to_host=False)
    for start, cnt in ((0, 50), (32768 - 20, 40), ((1 << 20) - 20, 57)):
        part = h.generate(cnt, philox_offset=11 + start * Z, to_host=False)
        assert torch.equal(big[start:start + cnt], part), "rows %d.." % start
    assert bool(torch.isfinite(big).all())
    h.close()


def test_inference_chains_written_out_equal_the_generic_launch(monkeypatch, kernel_path):
    """infer_seq.hip: encode / decode / generate of one-hidden-layer nets whose padded widths are 16 / 32 / 64 / 128 run a
    two-layer chain written out per shape (no loop over stage records, no switch over tile counts per 16-row block).
    Same arithmetic in the same order as the generic launch (SMRT_HIP_INFER_SEQ=0): bit-identical, ragged widths and a
    ragged row count included; nets outside the table (two hidden layers, 48-wide) take the generic launch either way."""
    if kernel_path != "fused":
        pytest.skip("the one-launch inference kernel belongs to the LDS-resident (fused) path")
    import torch
    n = 5003
    for D, hidden, Z in ((64, [128], 16), (16, [32], 8), (29, [60], 12), (128, [64], 32), (100, [64], 5), (40, [20, 12], 4)):
        rs = np.random.RandomState(D + Z)
        h = _handle(D, hidden, Z, seed=5, max_batch=64)
        h.set_params((rs.normal(size=h.n_params) * 0.2).astype(np.float32))
        X, Zs = rs.uniform(0, 1, size=(n, D)), rs.normal(size=(n, Z))
        outs = []
        for seq in ("1", "0"):
            monkeypatch.setenv("SMRT_HIP_INFER_SEQ", seq)
            mu, ls = h.encode(X)
            outs.append((mu, ls, h.decode(Zs), h.generate(n, philox_offset=8)))
        monkeypatch.delenv("SMRT_HIP_INFER_SEQ")
        for a, b, what in zip(outs[0], outs[1], ("mu", "log_sigma", "decode", "generate")):
            np.testing.assert_array_equal(a, b, err_msg="%s of %d-%s-%d" % (what, D, hidden, Z))
        assert np.isfinite(outs[0][2]).all()
        h.close()


# ---------------------------------------------------------------------------------------- persistent whole-epoch kernel
def _persist_pair(monkeypatch, D, hidden, Z, act, opt, max_batch, rs, lr=0.01, l2=1e-4, clip=True, seed=11):
    """(oracle, handle on the persistent path, handle on the two-kernel path) with the same injected weights."""
    o = _oracle(D, hidden, Z, act=act, opt=opt, lr=lr, l2=l2, clip=clip, seed=seed, rs=rs)
    hs = []
    for persist in ("1", "0"):
        monkeypatch.setenv("SMRT_HIP_PERSIST", persist)
        h = _handle(D, hidden, Z, act=act, opt=opt, lr=lr, l2=l2, clip=clip, seed=seed, max_batch=max_batch)
        h.set_params(flatten_params(o.params))
        hs.append(h)
    return o, hs[0], hs[1]


@pytest.mark.parametrize("D,hidden,Z,act,opt,n,B", [
    (64, [128], 16, "sigmoid", "rms_prop", 1000, 128),     # the bench net: 8 compute + 28 owner workgroups, ragged tail of 104
    (64, [128], 16, "sigmoid", "adam", 300, 128),          # Adam: beta powers kept per owner wave; tail of 44 rows (5 workgroups idle)
    (24, [16], 4, "tanh", "sgd", 150, 64),                 # batches 64, 64, 22: the last has rows in 2 of 4 workgroups
    (12, [7], 3, "relu", "adagrad", 37, 10),               # ONE compute workgroup, ragged widths (partial k-quads, bias quads)
    (30, [20, 12], 5, "elu", "adadelta", 700, 500),        # 32 compute workgroups (batch 500), two hidden layers
    (40, [72], 6, "identity", "proximal-adagrad", 130, 130),   # a single mini-batch per epoch
])
def test_persistent_epochs_match_two_kernel_path_and_oracle(monkeypatch, kernel_path, D, hidden, Z, act, opt, n, B):
    """One launch per epoch (persist_train.hip: compute workgroups + owner workgroups exchanging through L2) against the
    two-kernel step and against the oracle's fit: epoch costs, parameters, both optimizer slots, the step counter."""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    rs = np.random.RandomState(n)
    o, hp, h2 = _persist_pair(monkeypatch, D, hidden, Z, act, opt, B, rs)
    X = rs.uniform(0, 1, size=(n, D))
    nb = (n + B - 1) // B
    cp, _ = hp.train_epochs(X, B, 3)
    c2, _ = h2.train_epochs(X, B, 3)
    assert hp.last_epoch_launches == 2, hp.last_epoch_launches           # the persistent launch + the cost of the last batch
    assert h2.last_epoch_launches >= 2 * nb
    o.fit(X, n_epochs=3, batch_size=B)
    np.testing.assert_allclose(cp, np.asarray(o.epoch_costs_, np.float32), rtol=RTOL)
    np.testing.assert_allclose(cp, c2, rtol=2e-5)
    _close(hp.get_params(), flatten_params(o.params), "persistent params vs oracle")
    _close(hp.get_params(), h2.get_params(), "persistent vs two-kernel params", rtol=2e-5)
    for slot in (0, 1):
        _close(hp.get_opt_state(slot), h2.get_opt_state(slot), "slot %d" % slot, rtol=2e-5)
    assert hp.step_count == h2.step_count == 3 * nb
    # a second call continues from the device-resident state (step counter -> Philox stream, Adam powers), and single
    # steps through the two-kernel path pick up the weights the persistent launch left (flat buffer AND image twin)
    cp, _ = hp.train_epochs(X, B, 2)
    c2, _ = h2.train_epochs(X, B, 2)
    np.testing.assert_allclose(cp, c2, rtol=5e-5)
    Xs, eps = rs.uniform(0, 1, size=(min(B, 48), D)), rs.normal(size=(min(B, 48), Z))
    assert np.isclose(hp.train_step(Xs, eps), h2.train_step(Xs, eps), rtol=5e-5)
    cp, _ = hp.train_epochs(X, B, 1)
    c2, _ = h2.train_epochs(X, B, 1)
    np.testing.assert_allclose(cp, c2, rtol=1e-4)
    _close(hp.get_params(), h2.get_params(), "after epochs, a step, an epoch", rtol=1e-4)
    mu_p, ls_p = hp.encode(X[:40]); mu_2, ls_2 = h2.encode(X[:40])
    _close(mu_p, mu_2, "encode after persistent training", rtol=1e-4)
    hp.close(); h2.close()


@pytest.mark.parametrize("D,hidden,Z,form", [
    (64, [128], 16, 2),        # the bench net: 4 / 8 / 1 / 2 sixteen-wide blocks (features / hidden / latent / both heads)
    (16, [32], 8, 2),          # the smallest instantiation: 1 / 2 / 1 / 1
    (30, [60], 12, 2),         # ragged widths inside 32 / 64 / 16 / 32
    (128, [64], 32, 2),        # 8 / 4 / 2 / 4
    (100, [64], 16, 1),        # 112 features = 7 blocks: no instantiation, the generic kernel
    (30, [20, 12], 5, 1),      # two hidden layers: the generic kernel
])
def test_persistent_kernel_written_out_per_shape_equals_the_generic_one(monkeypatch, kernel_path, D, hidden, Z, form):
    """persist_seq.hip: for one-hidden-layer sigmoid VAEs of the common padded widths the stage sequence is written out (no
    loop, no switch, structure-specialised bodies).  Same arithmetic in the same order as the generic kernel: three epochs
    with a ragged tail leave bit-identical costs, parameters and optimizer slots; and the table of instantiations is what
    include/smrt_hip.h says it is."""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1")
    rs = np.random.RandomState(D + Z)
    n, B = 300, 128
    X = rs.uniform(0, 1, size=(n, D))
    o = _oracle(D, hidden, Z, act="sigmoid", opt="adam", lr=0.01, l2=1e-4, clip=True, seed=11, rs=rs)
    out = []
    for seq in (None, "0"):
        if seq is None:
            monkeypatch.delenv("SMRT_HIP_PERSIST_SEQ", raising=False)
        else:
            monkeypatch.setenv("SMRT_HIP_PERSIST_SEQ", seq)
        h = _handle(D, hidden, Z, act="sigmoid", opt="adam", lr=0.01, l2=1e-4, clip=True, seed=11, max_batch=B)
        h.set_params(flatten_params(o.params))
        costs, _ = h.train_epochs(X, B, 3)
        assert h.last_epoch_launches == 2
        assert h.persistent_form == (form if seq is None else 1)
        out.append((np.asarray(costs), h.get_params(), h.get_opt_state(0), h.get_opt_state(1)))
        h.close()
    for a, b, what in zip(out[0], out[1], ("costs", "parameters", "slot 0", "slot 1")):
        np.testing.assert_array_equal(a, b, err_msg=what)
    o.fit(X, n_epochs=3, batch_size=B)
    np.testing.assert_allclose(out[0][0], np.asarray(o.epoch_costs_, np.float32), rtol=RTOL)
    _close(out[0][1], flatten_params(o.params), "written-out kernel vs oracle")


def test_every_written_out_instantiation_equals_the_generic_kernel(monkeypatch, kernel_path):
    """All 36 instantiations of persist_seq.hip (features 16 / 32 / 64 / 128 x hidden 32 / 64 / 128 x latent <= 8 / 16 / 32), each at a
    ragged width inside its padded one: two epochs with a ragged tail, bit-identical to the generic persistent kernel.  Shapes
    whose weights do not fit in LDS are not on this path at all (persistent_form 0) and are skipped."""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1")
    n, B, ran = 150, 64, 0
    for D16 in (16, 32, 64, 128):
        for H16 in (32, 64, 128):
            for Z, zmax in ((8, 8), (16, 16), (32, 32)):
                D, H, Zr = D16 - 3, H16 - 5, zmax - (1 if zmax > 8 else 2)          # ragged inside the padded widths
                rs = np.random.RandomState(D16 + H16 + Z)
                X = rs.uniform(0, 1, size=(n, D))
                p0 = None
                out = []
                for seq in (None, "0"):
                    if seq is None:
                        monkeypatch.delenv("SMRT_HIP_PERSIST_SEQ", raising=False)
                    else:
                        monkeypatch.setenv("SMRT_HIP_PERSIST_SEQ", seq)
                    h = _handle(D, [H], Zr, act="sigmoid", opt="rms_prop", lr=0.01, l2=1e-4, clip=True, seed=3, max_batch=B)
                    if p0 is None:
                        p0 = (rs.normal(size=h.n_params) * 0.1).astype(np.float32)
                    h.set_params(p0)
                    form = h.persistent_form
                    if form == 0:
                        h.close()
                        break
                    assert form == (2 if seq is None else 1), (D, H, Zr, form)
                    costs, _ = h.train_epochs(X, B, 2)
                    assert h.last_epoch_launches == 2
                    out.append((np.asarray(costs), h.get_params(), h.get_opt_state(0)))
                    h.close()
                if len(out) == 2:
                    ran += 1
                    for a, b, what in zip(out[0], out[1], ("costs", "parameters", "slot 0")):
                        np.testing.assert_array_equal(a, b, err_msg="%s of %d-%d-%d" % (what, D, H, Zr))
                    assert np.isfinite(out[0][0]).all()
    assert ran >= 30, ran


def test_persistent_calls_of_growing_length_and_other_data(monkeypatch, kernel_path):
    """Workspace that outlives a call (the pinned per-epoch cost mirror, the once-per-call transposed copy of X): a second call
    with MORE epochs (the mirror is re-allocated), then one on a LARGER X (the copy is re-allocated), then a smaller one, each
    against the two-kernel path.  (Round 3: the mirror's growth path freed the transposed copy still in use.)"""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    rs = np.random.RandomState(77)
    D, hidden, Z, B = 64, [128], 16, 128
    o, hp, h2 = _persist_pair(monkeypatch, D, hidden, Z, "sigmoid", "rms_prop", B, rs)
    for n, epochs in ((300, 2), (300, 7), (1500, 3), (200, 9)):
        X = rs.uniform(0, 1, size=(n, D))
        cp, _ = hp.train_epochs(X, B, epochs)
        c2, _ = h2.train_epochs(X, B, epochs)
        assert hp.last_epoch_launches == 2 and len(cp) == epochs
        np.testing.assert_allclose(cp, c2, rtol=1e-4)
    _close(hp.get_params(), h2.get_params(), "after four calls", rtol=1e-4)
    import torch
    assert bool(torch.isfinite(torch.zeros(4, device="cuda")).all())          # no HIP error left behind for the next caller
    hp.close(); h2.close()


def test_persistent_early_stopping_and_ineligible_nets(monkeypatch, kernel_path):
    """Early stopping reads the cost after every persistent launch; dropout nets keep the two-kernel step."""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    rs = np.random.RandomState(9)
    D, hidden, Z, n, B = 16, [8], 2, 64, 32
    o, hp, h2 = _persist_pair(monkeypatch, D, hidden, Z, "sigmoid", "sgd", B, rs, lr=1e-4, seed=5)
    X = rs.uniform(0, 1, size=(n, D))
    o.fit(X, n_epochs=30, batch_size=B, early_stopping=True, min_change=0.05)
    costs, _ = hp.train_epochs(X, B, 30, early_stopping=True, min_change=0.05)
    assert len(costs) == len(o.epoch_costs_) < 30 and hp.last_epoch_launches == 2
    np.testing.assert_allclose(costs, np.array(o.epoch_costs_, np.float32), rtol=RTOL)
    hp.close(); h2.close()
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1")
    hd = _handle(D, hidden, Z, drop=0.5, max_batch=B)
    hd.set_params(flatten_params(o.params))
    hd.train_epochs(X, B, 2)
    assert hd.last_epoch_launches > 2
    hd.close()


def test_persistent_timeout_is_sticky_and_falls_back(monkeypatch, kernel_path):
    """A persistent launch that gives up (a peer never became resident) raises an error word.  It must survive the epochs
    queued behind it (the per-launch memset used to clear it), the launches behind it must leave untouched what
    the failed epoch found, and smrt_vae_train_epochs must then run that epoch and the rest on the two-launch step.
    The test hook raises the word in front of epoch 1's launch, exactly what a time-out in that launch leaves behind."""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    from smrt_amd._hip import HipLibraryError
    rs = np.random.RandomState(21)
    D, hidden, Z, n, B = 64, [128], 16, 600, 128
    o, hp, h2 = _persist_pair(monkeypatch, D, hidden, Z, "sigmoid", "rms_prop", B, rs)
    X = rs.uniform(0, 1, size=(n, D))
    nb = (n + B - 1) // B
    c2, _ = h2.train_epochs(X, B, 4)
    # (1) reported, not lost: epochs 2 and 3 run behind the failed one and must not clear the word
    monkeypatch.setenv("SMRT_HIP_PERSIST_FAULT_EPOCH", "1")
    monkeypatch.setenv("SMRT_HIP_PERSIST_NO_FALLBACK", "1")
    with pytest.raises(HipLibraryError, match=r"\(-6\)"):
        hp.train_epochs(X, B, 4)
    # the handle is where epoch 0 left it: one complete epoch, nothing of the failed ones
    assert hp.step_count == nb
    monkeypatch.delenv("SMRT_HIP_PERSIST_FAULT_EPOCH")
    monkeypatch.delenv("SMRT_HIP_PERSIST_NO_FALLBACK")
    c_rest, _ = hp.train_epochs(X, B, 3)
    np.testing.assert_allclose(c_rest, c2[1:], rtol=5e-5)
    _close(hp.get_params(), h2.get_params(), "resumed after a reported time-out", rtol=5e-5)
    assert hp.step_count == h2.step_count == 4 * nb
    hp.close(); h2.close()
    # (2) recovered: same fault, fallback allowed -> the call succeeds, results as the two-launch path's
    o, hp, h2 = _persist_pair(monkeypatch, D, hidden, Z, "sigmoid", "rms_prop", B, rs)
    c2, _ = h2.train_epochs(X, B, 4)
    monkeypatch.setenv("SMRT_HIP_PERSIST_FAULT_EPOCH", "2")
    cp, ms = hp.train_epochs(X, B, 4)
    monkeypatch.delenv("SMRT_HIP_PERSIST_FAULT_EPOCH")
    assert len(cp) == 4 and hp.last_epoch_launches >= 2 * nb          # the last epochs ran as two launches per mini-batch
    np.testing.assert_allclose(cp, c2, rtol=5e-5)
    _close(hp.get_params(), h2.get_params(), "after the fallback", rtol=5e-5)
    for slot in (0, 1):
        _close(hp.get_opt_state(slot), h2.get_opt_state(slot), "slot %d after the fallback" % slot, rtol=5e-5)
    assert hp.step_count == 4 * nb
    # with early stopping the word is read after every epoch
    monkeypatch.setenv("SMRT_HIP_PERSIST_FAULT_EPOCH", "0")
    ce, _ = hp.train_epochs(X, B, 2, early_stopping=True, min_change=1e-9)
    monkeypatch.delenv("SMRT_HIP_PERSIST_FAULT_EPOCH")
    c2e, _ = h2.train_epochs(X, B, 2, early_stopping=True, min_change=1e-9)
    np.testing.assert_allclose(ce, c2e, rtol=1e-4)
    hp.close(); h2.close()


def test_persistent_budget_is_per_device_and_plannable(monkeypatch, kernel_path):
    """The budget of co-resident persistent workgroups: derived from the device, overridable, and a handle can be told
    which path to take so that concurrent fits do not race for it (fit_group plans in class order)."""
    if kernel_path != "fused":
        pytest.skip("the persistent kernel belongs to the LDS-resident (fused) path")
    from smrt_amd import _hip
    monkeypatch.delenv("SMRT_HIP_PERSIST_BUDGET", raising=False)
    monkeypatch.setenv("SMRT_HIP_PERSIST", "1")
    assert _hip.persistent_budget(0) == 160                         # 5/8 of 256 CUs, one rank on the GPU
    rs = np.random.RandomState(3)
    D, hidden, Z, n, B = 64, [128], 16, 300, 128
    X = rs.uniform(0, 1, size=(n, D))
    nb = (n + B - 1) // B
    h = _handle(D, hidden, Z, max_batch=B)
    need = h.persistent_workgroups(B)
    assert need == 8 + 28 and h.persistent_workgroups(513) == 0
    h.train_epochs(X, B, 2)
    assert h.last_epoch_launches == 2
    monkeypatch.setenv("SMRT_HIP_PERSIST_BUDGET", str(need - 1))    # does not fit: the two-launch step
    assert _hip.persistent_budget(0) == need - 1
    h.train_epochs(X, B, 2)
    assert h.last_epoch_launches >= 2 * nb
    h.set_persistent(1)                                             # planned by the caller: granted whatever the budget says
    h.train_epochs(X, B, 2)
    assert h.last_epoch_launches == 2
    monkeypatch.delenv("SMRT_HIP_PERSIST_BUDGET")
    h.set_persistent(0)
    h.train_epochs(X, B, 2)
    assert h.last_epoch_launches >= 2 * nb
    h.close()
