"""Where a generate call of the bench net goes: Python wrapper, C call, launch + completion, kernel (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smrt_amd import _hip
h = _hip.VaeHandle(n_features=64, hidden=[128], n_latent=16, activation="sigmoid", optimizer="rms_prop",
                   learning_rate=0.05, l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10,
                   dropout=1.0, seed=42, max_batch=128, device=0)
def t(fn, reps=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps): fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps
print("generate 16 rows        %.1f us" % t(lambda: h.generate(16, to_host=False)))
print("generate 176470 rows    %.1f us" % t(lambda: h.generate(176470, to_host=False)))
print("torch.empty             %.1f us" % t(lambda: torch.empty((176470, 64), dtype=torch.float32, device="cuda")))
print("sync_torch_stream       %.1f us" % t(lambda: _hip.sync_torch_stream(h.device)))
out = torch.empty((176470, 64), dtype=torch.float32, device="cuda")
lib = h._lib; hp = h._live()
print("raw C call 16 rows      %.1f us" % t(lambda: lib.smrt_vae_generate(hp, 16, 0, 0.0, 1.0, out.data_ptr(), 64)))
print("raw C call 176470 rows  %.1f us" % t(lambda: lib.smrt_vae_generate(hp, 176470, 0, 0.0, 1.0, out.data_ptr(), 64)))
h.close()
