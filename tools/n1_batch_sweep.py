"""N1 = 784-450-(10,10)-450-784 training throughput over mini-batch sizes 2048 .. 65536 (adam), per-launch averages by kind:
what one rank of a data-parallel C4 run sees at 65536 / world rows per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import bench_configs as bc
for b in (2048, 4096, 8192, 16384, 32768, 65536):
    r = bc.case_train("n1_b%d" % b, "N1", max(4 * b, 65536), b, "adam", 3)
    print(b, round(r["tflops"], 1), round(r["ms_per_epoch_wall"] / (max(4 * b, 65536) // b) * 1e3, 1), "us/step",
          {k: round(v["ms_per_epoch"] / v["launches"] * 1e3, 1) for k, v in r["kernels"].items()})
