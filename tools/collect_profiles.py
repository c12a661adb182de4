#!/usr/bin/env python
"""Copy the round's rocprofv3 evidence (gpurun_out/r03/, written by tools/profile_round.sh on the GPU box) into profiles/:
kernel-stats CSVs as they are, --pmc passes summarised per kernel (counter sums per launch), the bench line and the
per-configuration lines.  python tools/collect_profiles.py [tag]   (tag defaults to r03)"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = os.path.join(ROOT, "gpurun_out", "r03"), os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].replace("void ", "").replace("smrt::", "").strip()


def stats(sub, out):
    f = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, out)))


def pmc(sub, out, command, keep):
    res = {"command": command, "units": "counter sums per launch, over the launch's waves", "kernels": {}}
    for p in "ab":
        tot, seen = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(set)
        for f in glob.glob(os.path.join(src, sub, p, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not any(k.startswith(w) for w in keep):
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
        for k in tot:
            rec = res["kernels"].setdefault(k, {})
            rec["launches_" + p] = len(seen[k])
            for c, v in tot[k].items():
                rec[c] = round(v / max(1, len(seen[k])), 1)
    for k, rec in res["kernels"].items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("SQ_BUSY_CYCLES"):
            rec["mfma_busy_over_wave_cycles"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, rec.get("SQ_WAVE_CYCLES", 0.0)), 4)
    json.dump(res, open(os.path.join(dst, "%s_%s_pmc_sq.json" % (tag, out)), "w"), indent=1)


for sub, out in (("n1_b128", "n1_b128"), ("n1_b256", "n1_b256"), ("c3", "c3"), ("n1_b8192", "n1_b8192"), ("n1_b65536", "n1_b65536")):
    stats(sub, out)
P = "rocprofv3 --pmc <8 SQ counters> (two passes) -- python3 "
pmc("sq_n1_b128", "n1_b128", P + "tools/bench_configs.py --cases train_n1_b128", ("small_", "mid_", "finalize"))
pmc("sq_n1_b65536", "n1_b65536", P + "tools/n1_large.py 65536", ("dw_kernel", "gemm_kernel", "small_gemm", "optimizer"))
pmc("sq_infer", "infer", P + "tools/infer_only.py", ("fused_infer",))
f = os.path.join(src, "pmc_persist_r03", "summary.json")
if os.path.exists(f):
    shutil.copy(f, os.path.join(dst, tag + "_persist_pmc_sq.json"))
for name, out in (("bench_line.json", "_bench_c2_line.json"), ("configs.jsonl", "_configs.jsonl")):
    f = os.path.join(src, name)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, tag + out))
# kernel stats + HBM traffic of the bench command (tools/pmc_traffic.py reads gpurun_out/prof_bench)
pb = os.path.join(ROOT, "gpurun_out", "prof_bench")
if os.path.isdir(os.path.join(src, "bench")):
    if os.path.isdir(pb):
        shutil.rmtree(pb)
    shutil.copytree(os.path.join(src, "bench"), pb)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), tag + "_bench_c2"])
print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(dst, tag + "_*"))))
