"""Phase timing of smrt_balance on config C2 (1M x 64 fp32, 1:50; SMRT_TIMING=1 prints one line per phase on stderr)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SMRT_TIMING"] = "1"
import numpy as np
from smrt_amd.balance import smrt_balance
from bench import tabular_c2
X, y = tabular_c2()
kw = dict(n_hidden=128, n_latent_factors=16, balance_ratio=0.2, random_state=42)
smrt_balance(X[:50000], y[:50000], n_epochs=1, **kw)
for _ in range(2):
    sys.stderr.write("----\n")
    t0 = time.perf_counter(); smrt_balance(X, y, **kw); sys.stderr.write("total %.1f ms\n" % (1e3 * (time.perf_counter() - t0)))
