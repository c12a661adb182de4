"""A/B of two library builds on ONE box: N1 = 784-450-(10,10)-450-784 training at the reference's batch sizes, each arm in a
process of its own (SMRT_HIP_LIB), arms alternated, several repetitions.  Usage: python tools/ab_n1.py [libB] [reps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib_b = sys.argv[1] if len(sys.argv) > 1 else "smrt_amd/_lib/ab/libsmrt_hip.so"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cases = os.environ.get("AB_CASES", "train_n1_b128,train_n1_b256")
res = {}
for rep in range(reps):
    for arm, lib in (("A", None), ("B", lib_b)):
        env = dict(os.environ)
        if lib:
            env["SMRT_HIP_LIB"] = os.path.join(ROOT, lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "--cases", cases],
                             env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            r = json.loads(line)
            if "batch" in r:
                v = r["ms_per_epoch_wall"] * 1e3 / (-(-r["rows"] // r["batch"]))
            else:
                v = 1e3 * r["wall_s_median"]
            res.setdefault((r["case"], arm), []).append(v)
for (case, arm), v in sorted(res.items()):
    print(case, arm, "median %.1f" % sorted(v)[len(v) // 2], ["%.1f" % x for x in v])
