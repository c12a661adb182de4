#!/usr/bin/env python
"""Achievable peaks and latency floors on this MI355X, through the library's diagnostic entry points.

    python tools/peaks.py            -> one JSON object: fp32 MFMA TFLOP/s, HBM copy/read GB/s, and the cost of a
                                        mini-batch step that does no arithmetic (captured chain of dependent kernels)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smrt_amd import _hip  # noqa: E402


def main():
    out = {"mfma_f32_tflops": _hip.diag_mfma_peak(0), "mfma_f32_nominal_tflops": 157.3}
    c, r = _hip.diag_hbm_bandwidth(0, 1 << 30)
    out.update({"hbm_copy_gbs": c, "hbm_read_gbs": r, "hbm_nominal_gbs": 8000.0})
    floors = {}
    for name, k, wgs, thr in [("c2_fused_2_kernels_8wg", 2, 8, 512), ("one_kernel_8wg", 1, 8, 512),
                              ("n1_per_layer_11_kernels_120wg", 11, 120, 256), ("one_kernel_1024wg", 1, 1024, 256)]:
        floors[name] = {"kernels_per_step": k, "workgroups": wgs, "threads": thr,
                        "us_per_step": _hip.diag_step_floor(0, k, 154, wgs, thr)}
    out["step_floor"] = floors
    print(json.dumps(out))


if __name__ == "__main__":
    main()
