import numpy as np, time, sys
sys.path.insert(0, ".")
from smrt_amd import _hip
from sklearn.utils.validation import check_array
X = np.random.RandomState(0).rand(1000000, 64).astype(np.float32)
_hip.all_finite(X)
for nt in (1, 4, 16):
    t0 = time.perf_counter(); r = _hip.all_finite(X, nt); print(nt, r, "all_finite ms", 1e3 * (time.perf_counter() - t0))
t0 = time.perf_counter(); check_array(X, dtype=[np.float64, np.float32], copy=False, order="C"); print("sklearn finite ms", 1e3 * (time.perf_counter() - t0))
