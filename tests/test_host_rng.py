"""smrt_rng_normal / smrt_rng_permutation (smrt_amd/csrc/host_rng.cpp) against numpy's own legacy RandomState: the same
values bit for bit and the same generator state afterwards -- whatever the state they start from (mid-block position, a
kept normal from an odd earlier draw), for odd and even counts, with loc / scale, and through the thread pool."""
import numpy as np
import pytest

from smrt_amd import _hip


def _same_state(a, b):
    sa, sb = a.get_state(legacy=True), b.get_state(legacy=True)
    return (sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2] == sb[2] and sa[3] == sb[3]
            and (sa[4] == sb[4] or (np.isnan(sa[4]) and np.isnan(sb[4]))))


@pytest.mark.parametrize("seed,warm,n,loc,scale", [
    (42, 0, 176470 * 16, 0.0, 1.0),          # C2's latent draw, fresh generator
    (0, 7, 100001, 0.0, 1.0),                # odd count after an odd warm-up: starts with the kept normal, ends keeping one
    (123456789, 625, 65536, -1.5, 0.25),     # mid-block start, loc / scale
    (7, 3, 4097, 2.0, 0.0),                  # scale 0
])
def test_normal_is_numpy_legacy_normal_bit_for_bit(seed, warm, n, loc, scale):
    a, b = np.random.RandomState(seed), np.random.RandomState(seed)
    if warm:
        a.normal(size=warm); b.normal(size=warm)
    want = a.normal(loc, scale, n)
    got = _hip.rng_normal(b, n, loc, scale)
    assert got.dtype == np.float64 and got.shape == (n,)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert _same_state(a, b)
    assert np.array_equal(a.normal(size=5), b.normal(size=5)) and a.randint(1 << 30) == b.randint(1 << 30)


def test_normal_shapes_threads_and_the_numpy_fallbacks():
    a, b = np.random.RandomState(5), np.random.RandomState(5)
    want = a.normal(size=(5000, 3))
    for nt in (1, 3, 16):
        c = np.random.RandomState(5)
        assert np.array_equal(_hip.rng_normal(c, (5000, 3), n_threads=nt), want)
        assert _same_state(a, c)
    b.normal(size=(5000, 3))
    # small draws and array-valued loc go to numpy itself: still the same stream
    assert np.array_equal(_hip.rng_normal(b, 10), a.normal(size=10))
    assert np.array_equal(_hip.rng_normal(b, (4, 2), loc=np.array([0.0, 1.0])), a.normal(np.array([0.0, 1.0]), 1.0, (4, 2)))
    assert _same_state(a, b)


@pytest.mark.parametrize("seed,warm,n", [(42, 0, 1176470), (1, 11, 4096), (99, 700, 100003)])
def test_permutation_is_numpy_legacy_permutation_bit_for_bit(seed, warm, n):
    a, b = np.random.RandomState(seed), np.random.RandomState(seed)
    if warm:
        a.normal(size=warm); b.normal(size=warm)          # (odd warm-ups leave a kept normal: a shuffle must not touch it)
    want = a.permutation(np.arange(n))
    got = _hip.rng_permutation(b, n)
    assert got.dtype == np.int64 and np.array_equal(got, want)
    assert _same_state(a, b)
    assert np.array_equal(a.normal(size=3), b.normal(size=3))


def test_the_sequence_smrt_balance_draws():
    """choice (numpy) -> normal -> normal -> permutation on one stream, as smrt_balance interleaves them."""
    a, b = np.random.RandomState(42), np.random.RandomState(42)
    wa = [a.choice(np.arange(5000), size=7000, replace=True), a.normal(size=(7001, 10)), a.normal(size=(6999, 10)),
          a.permutation(np.arange(90000))]
    wb = [b.choice(np.arange(5000), size=7000, replace=True), _hip.rng_normal(b, (7001, 10)), _hip.rng_normal(b, (6999, 10)),
          _hip.rng_permutation(b, 90000)]
    for x, y in zip(wa, wb):
        assert np.array_equal(x, y)
    assert _same_state(a, b)


def test_work_arrays_are_reused_across_calls_of_other_sizes_and_threads():
    """The work arrays of both entry points live per calling thread and only grow: calls of growing, shrinking and odd
    sizes on one thread, and the same sequence on two other threads at once, must each stay numpy's stream bit for bit."""
    from concurrent.futures import ThreadPoolExecutor

    def sequence(seed):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        for n_norm, n_perm in ((4097, 5000), (300001, 70000), (8193, 2), (65537, 250000), (4099, 4097)):
            want = a.normal(size=n_norm)
            got = _hip.rng_normal(b, n_norm)
            if not np.array_equal(got.view(np.uint64), want.view(np.uint64)):
                return False
            if not np.array_equal(a.permutation(n_perm), _hip.rng_permutation(b, n_perm)):
                return False
            if not _same_state(a, b):
                return False
        return True

    assert sequence(11)
    with ThreadPoolExecutor(max_workers=2) as pool:
        assert all(pool.map(sequence, (12, 13)))
