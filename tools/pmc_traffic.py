#!/usr/bin/env python
"""Turn the rocprofv3 outputs of tools/profile_bench.sh (gpurun_out/prof_bench/) into committed summaries:

    profiles/<tag>_kernel_stats.csv      copy of the --kernel-trace --stats summary
    profiles/<tag>_traffic.json          per kernel: launches, average duration, HBM bytes per launch

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, both counters in KiB as rocprofv3 reports them and
collected in separate passes; the factor 2 is the gfx950 correction of MI355X_MICROARCH.md ("FETCH_SIZE
reports exactly 1/2 of the bytes of a wide coalesced streaming read").  The summary is stamped with a hash of the
kernel sources (bench.csrc_hash); bench.py reports roofline.traffic only from a summary whose stamp matches the
sources it runs, and `traffic: null, traffic_stale: true` otherwise.  Run it right after tools/profile_bench.sh,
before editing smrt_amd/csrc/.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").replace("smrt::", "").strip()


def per_kernel_counter(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            n[k] += 1
    return tot, n


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01_bench_c2"
    src = os.path.join(ROOT, "gpurun_out", "prof_bench")
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    stats = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv")))}
    fetch, nf = per_kernel_counter(os.path.join(src, "fetch", "bench_counter_collection.csv"), "FETCH_SIZE")
    write, nw = per_kernel_counter(os.path.join(src, "write", "bench_counter_collection.csv"), "WRITE_SIZE")
    sys.path.insert(0, ROOT)
    from bench import csrc_hash
    out = {"csrc_hash": csrc_hash(), "command": "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-balance",
           "units": "bytes per launch; FETCH_SIZE/WRITE_SIZE are KiB, FETCH_SIZE doubled (gfx950 correction)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if k not in stats or not (k.startswith("fused") or k.startswith("persist") or k.startswith("gemm") or k.startswith("dw_") or
                                  k.startswith("optimizer") or k.startswith("finalize")):
            continue
        f = 1024.0 * fetch[k] / max(nf[k], 1)
        w = 1024.0 * write[k] / max(nw[k], 1)
        out["kernels"][k] = {"launches": int(stats[k]["Calls"]), "avg_ns": float(stats[k]["AverageNs"]),
                             "fetch_size_bytes_raw": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2.0 * f + w}
    json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
