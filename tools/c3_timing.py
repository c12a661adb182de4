"""Phase timing of smrt_balance on config C3 (SMRT_TIMING=1 prints one line per phase on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SMRT_TIMING"] = "1"
import numpy as np
from smrt_amd.balance import smrt_balance
rs = np.random.RandomState(42)
Xs, ys = [], []
for label, n in [(0, 50000)] + [(k, 5000) for k in range(1, 9)]:
    proto, mask = rs.uniform(size=784), rs.uniform(size=784) < 0.2
    Xs.append((np.clip(proto + 0.15 * rs.normal(size=(n, 784)), 0, 1) * mask).astype(np.float32))
    ys.append(np.full(n, label))
X, y = np.vstack(Xs), np.concatenate(ys)
p = rs.permutation(len(y)); X, y = X[p], y[p]
kw = dict(n_hidden=450, n_latent_factors=10, balance_ratio=0.5, random_state=42)
smrt_balance(X[:9000], y[:9000], n_epochs=1, **kw)
for _ in range(2):
    sys.stderr.write("----\n")
    t0 = time.perf_counter(); smrt_balance(X, y, **kw); sys.stderr.write("total %.1f ms\n" % (1e3 * (time.perf_counter() - t0)))
