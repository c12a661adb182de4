import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from smrt_amd import _hip
D, hidden, Z = 64, [128], 16
h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer="rms_prop", learning_rate=0.05, l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=42, max_batch=128, device=0)
n = 176470
for rep in range(3):
    for seq in ("1", "0"):
        os.environ["SMRT_HIP_INFER_SEQ"] = seq
        for i in range(20): h.generate(n, philox_offset=i * n * Z, to_host=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(300): h.generate(n, philox_offset=i * n * Z, to_host=False)
        torch.cuda.synchronize()
        print("seq", seq, "generate %.4f ms" % (1e3 * (time.perf_counter() - t0) / 300), flush=True)
