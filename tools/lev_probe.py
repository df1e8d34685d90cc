import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flashdeconv_amd.utils import genes
X = np.random.RandomState(0).randn(30, 2000)
for _ in range(5):
    t0 = time.perf_counter(); genes.compute_leverage_scores(X); print("ms", round((time.perf_counter() - t0) * 1e3, 2))
