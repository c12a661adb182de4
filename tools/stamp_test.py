import numpy as np, sys
sys.path.insert(0, '.')
from smrt_amd import _hip
h = _hip.VaeHandle(n_features=64, hidden=[128], n_latent=16, activation="sigmoid", optimizer="rms_prop", learning_rate=0.05, l2_penalty=1e-4, clip=True, gclip_min=-5., gclip_max=5., eps=1e-10, dropout=1.0, seed=42, max_batch=128, device=0)
rs = np.random.RandomState(0)
h.set_params(rs.normal(size=h.n_params).astype(np.float32)*0.1)
X = rs.uniform(size=(128,64))
for i in range(6): h.train_step(X)
