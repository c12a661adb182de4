import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["SMRT_TIMING"] = "1"
import numpy as np
import bench
from smrt_amd.balance import smrt_balance
X, y = bench.tabular_c2()
kw = dict(n_hidden=[128], n_latent_factors=16, balance_ratio=0.2, random_state=42)
smrt_balance(X[:50000], y[:50000], **kw)
for _ in range(3):
    sys.stderr.write("----\n")
    t0 = time.perf_counter(); smrt_balance(X, y, **kw); sys.stderr.write("total %.1f ms\n" % (1e3 * (time.perf_counter() - t0)))
