#!/usr/bin/env python
"""Host-side cost of preparing ONE class fit: C3's (5000 x 784 rows, N1) by default, C5's with `n5` (19608 x 256, N5):
where fit_group's `prepare` / the time in front of a fit's first epoch goes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sklearn.utils.validation import check_array
from smrt_amd import _hip
from smrt_amd.autoencode.topography import VAETopography
from smrt_amd.utils import get_random_state

N5 = len(sys.argv) > 1 and sys.argv[1] == "n5"
D, HID, ZL, ROWS, MB = (256, [1024] * 4, 64, 19608, 4096) if N5 else (784, [450], 10, 5000, 128)
X = np.random.RandomState(0).rand(ROWS, D).astype(np.float32)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for rep in range(3):
    t = [time.perf_counter()]
    Xc = check_array(X, accept_sparse=False, ensure_2d=True, dtype=[np.float64, np.float32]); t.append(time.perf_counter())
    rs = get_random_state(42)
    topo = VAETopography(n_hidden=HID, input_shape=D, activation="sigmoid", n_latent_factors=ZL, random_state=rs); t.append(time.perf_counter())
    init = topo.initial_parameters(); t.append(time.perf_counter())
    flat = VAETopography.join_flat(init); t.append(time.perf_counter())
    h = _hip.VaeHandle(n_features=D, hidden=HID, n_latent=ZL, activation="sigmoid", optimizer="rms_prop", learning_rate=0.005,
                       l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=42, max_batch=MB, device=0)
    t.append(time.perf_counter())
    h.set_params(flat); t.append(time.perf_counter())
    Xd = _hip.to_device_padded(Xc, h.device); torch.cuda.synchronize(); t.append(time.perf_counter())
    h.close(); t.append(time.perf_counter())
    names = ["check_array", "topography", "initial_parameters", "join_flat", "VaeHandle", "set_params", "to_device_padded", "close"]
    print("  ".join("%s %.2f" % (n, 1e3 * (b - a)) for n, a, b in zip(names, t[:-1], t[1:])), "ms")
