import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import oracle.matsed_oracle as O
from transformer4sed_amd.filter import median_filter_torch, median_filter_scipy, max_filter_scipy
rng = np.random.RandomState(0)
for T in (1, 5, 20, 64, 1000):
    x = rng.rand(2, T, 10).astype(np.float32)
    for sizes in ([33, 129, 1, 3, 2, 4, 7, 65, 31, 128], [1] * 10):
        a = median_filter_torch(torch.from_numpy(x).cuda(), sizes).cpu().numpy()
        b = O.median_filter_torchpath(x, sizes)
        c = median_filter_scipy(torch.from_numpy(x).cuda(), sizes).cpu().numpy()
        d = O.median_filter_scipypath(x, sizes)
        print(T, sizes[:3], "torch-path equal", np.array_equal(a, b), " scipy-path equal", np.array_equal(c, d))
try:
    e = median_filter_torch(torch.zeros(0, 1000, 10).cuda(), [3] * 10)
    print("empty batch ok", tuple(e.shape))
except Exception as ex:
    print("empty batch raises", repr(ex)[:200])
