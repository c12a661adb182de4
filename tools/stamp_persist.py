"""In-kernel phase stamps of the persistent epoch kernel on the bench net 64-128-(16,16)-128-64 at batch 128 (diagnostic
build: make -C smrt_amd/csrc stamps): SMRT_HIP_LIB=smrt_amd/_lib/stamps/libsmrt_hip.so python tools/stamp_persist.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SMRT_HIP_PERSIST", "1")
from smrt_amd import _hip
D, hidden, Z, B = 64, [128], 16, 128
rs = np.random.RandomState(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X = (1.0 / (1.0 + np.exp(-(rs.normal(size=D) + rs.normal(size=(n, D)))))).astype(np.float32)
h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer="rms_prop", learning_rate=0.01,
                   l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=42, max_batch=B, device=0)
h.set_params((rs.normal(size=h.n_params) * 0.1).astype(np.float32))
Xd = _hip.to_device_padded(X, h.device)
costs, ms = h.train_epochs(Xd, B, 6)
print("epoch ms", ms, "us per mini-batch", 1e3 * float(np.median(ms)) / (n // B), "launches/epoch", h.last_epoch_launches)
