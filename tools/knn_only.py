"""One exact k-NN search over a C2-sized minority class (19 608 x 64 fp64, k = 6): the launch rocprofv3 is pointed at."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from smrt_amd import _hip  # noqa: E402

n, d, k = (int(a) for a in (sys.argv[1:4] + ["19608", "64", "6"][len(sys.argv) - 1:]))
X = torch.from_numpy(np.random.RandomState(0).normal(size=(n, d))).cuda()
for _ in range(3):
    idx = _hip.knn_f64(X, k, to_host=False)
torch.cuda.synchronize()
print("ok", tuple(idx.shape))
