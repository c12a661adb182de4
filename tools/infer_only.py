#!/usr/bin/env python
"""Inference launches of the bench net alone, for rocprofv3 (--kernel-trace --stats / --pmc):
generate 176 470 rows (C2's synthetic rows), decode and encode 1 M rows of N2 = 64-128-(16,16)-128-64.
Prints wall-clock per call (host side, synchronised) as one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from smrt_amd import _hip  # noqa: E402


def main():
    D, hidden, Z = 64, [128], 16
    h = _hip.VaeHandle(n_features=D, hidden=hidden, n_latent=Z, activation="sigmoid", optimizer="rms_prop",
                       learning_rate=0.05, l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10,
                       dropout=1.0, seed=42, max_batch=128, device=0)
    n_gen, n_big, reps = 176470, 1 << 20, 10
    X = torch.rand(n_big, D, device="cuda")
    Zs = torch.randn(n_big, Z, device="cuda")
    out = {}
    for name, fn in (("generate_176470", lambda i: h.generate(n_gen, philox_offset=i * n_gen * Z, to_host=False)),
                     ("decode_1M", lambda i: h.decode(Zs, to_host=False)),
                     ("encode_1M", lambda i: h.encode(X, to_host=False))):
        fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            fn(i)
        torch.cuda.synchronize()
        out[name + "_ms"] = 1e3 * (time.perf_counter() - t0) / reps
    out["decode_1M_GBps"] = n_big * 4 * (Z + D) / (out["decode_1M_ms"] * 1e-3) / 1e9
    out["encode_1M_GBps"] = n_big * 4 * (D + 2 * Z) / (out["encode_1M_ms"] * 1e-3) / 1e9
    h.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
