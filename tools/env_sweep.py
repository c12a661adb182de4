"""Sweep one environment variable over values, each in a process of its own on ONE box (tools/bench_configs.py cases in
AB_CASES): python tools/_sweep.py SMRT_HIP_TILE_FILL 120,200,300"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals = sys.argv[1], sys.argv[2].split(",")
cases = os.environ.get("AB_CASES", "train_n1_b128,train_n1_b256")
for rep in range(2):
    for v in vals:
        env = dict(os.environ); env[var] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_configs.py"), "--cases", cases], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            r = json.loads(line)
            if "batch" in r:
                print(var, v, r["case"], "%.1f us/step" % (r["ms_per_epoch_wall"] * 1e3 / (-(-r["rows"] // r["batch"]))), flush=True)
            else:
                print(var, v, r["case"], r["wall_s_all"], flush=True)
