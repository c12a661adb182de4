"""Persistent whole-epoch kernel vs the two-kernel step and the oracle on the bench net (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.vae_oracle import OracleVAE, flatten_params
from smrt_amd import _hip

D, hidden, Z, B = 64, [128], 16, 128
rs = np.random.RandomState(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
opt = sys.argv[2] if len(sys.argv) > 2 else "rms_prop"
X = (1.0 / (1.0 + np.exp(-(rs.normal(size=D) + rs.normal(size=(n, D)))))).astype(np.float32)
o = OracleVAE(D, hidden, Z, seed=42, learning_function=opt, learning_rate=0.01)
p0 = flatten_params(o.params).copy()
res = {}
for persist in ("1", "0"):
    os.environ["SMRT_HIP_PERSIST"] = persist
    h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer=opt, learning_rate=0.01,
                       l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=42,
                       max_batch=B, device=0)
    h.set_params(p0)
    Xd = _hip.to_device_padded(X, h.device)
    costs, ms = h.train_epochs(Xd, B, 3)
    print("persist", persist, "costs", costs, "epoch ms", ms, "launches/epoch", h.last_epoch_launches, flush=True)
    res[persist] = (costs, h.get_params(), h.get_opt_state(0), h.get_opt_state(1))
    t0 = time.perf_counter(); c2, ms2 = h.train_epochs(Xd, B, 20); dt = time.perf_counter() - t0
    print("   20 more epochs: %.3f ms/epoch wall, device %.3f ms/epoch" % (1e3 * dt / 20, float(np.mean(ms2))), flush=True)
    h.close()
o.fit(X, n_epochs=3, batch_size=B)
po = flatten_params(o.params)
for k, (costs, p, a, b) in res.items():
    print("persist", k, "vs oracle: cost rel", np.max(np.abs(costs - np.array(o.epoch_costs_)) / np.abs(o.epoch_costs_)),
          "param abs", np.abs(p - po).max(), "scale", np.abs(po).max())
print("persist vs two-kernel: param abs", np.abs(res["1"][1] - res["0"][1]).max(), "slot0", np.abs(res["1"][2] - res["0"][2]).max(),
      "slot1", np.abs(res["1"][3] - res["0"][3]).max())
