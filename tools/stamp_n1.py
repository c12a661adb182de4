"""In-kernel phase stamps of the small-batch step of N1 = 784-450-(10,10)-450-784 (diagnostic build: make -C smrt_amd/csrc stamps):
SMRT_HIP_LIB=smrt_amd/_lib/stamps/libsmrt_hip.so python tools/stamp_n1.py [batch]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from smrt_amd import _hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h = _hip.VaeHandle(n_features=784, hidden=[450], n_latent=10, activation="sigmoid", optimizer="rms_prop", learning_rate=0.005,
                   l2_penalty=1e-4, clip=True, gclip_min=-5., gclip_max=5., eps=1e-10, dropout=1.0, seed=42, max_batch=B, device=0)
rs = np.random.RandomState(0)
h.set_params(rs.normal(size=h.n_params).astype(np.float32) * 0.05)
X = _hip.to_device_padded(rs.uniform(size=(B, 784)), h.device)
for i in range(8):
    h.train_step(X)
