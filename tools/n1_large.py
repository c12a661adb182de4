"""N1 = 784-450-(10,10)-450-784 at large mini-batches (what a data-parallel C4 rank trains per step), per-launch averages by kind."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import bench_configs as bc
for b in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("4096", "8192", "65536"))]:
    rows = max(4 * b, 65536)
    r = bc.case_train("n1_b%d" % b, "N1", rows, b, "adam", 3)
    print(os.environ.get("TAG", ""), b, round(r["tflops"], 1), "TF", round(r["ms_per_epoch_wall"] / (rows // b) * 1e3, 1), "us/step",
          {k: round(v["ms_per_epoch"] / v["launches"] * 1e3, 1) for k, v in r["kernels"].items()}, flush=True)
