#!/usr/bin/env python3
"""How far the HVG set of a FLOAT32 count matrix can differ from the reference's (numpy forms the per-gene statistics in float32, the
device in float64 from the float32 values): overlap of the two sets at a realistic size, with the device's moments used as they are and
rounded to float32 first.  usage: python tools/hvg_f32_probe.py [n] [G] [n_top]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import fdx_oracle as orc  # noqa: E402
from flashdeconv_amd.utils import genes as g  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
n_top = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rs = np.random.RandomState(0)
K = 12
X = np.exp(rs.randn(K, G) * 1.0) * rs.gamma(0.3, 1.0, G)
B = rs.dirichlet(np.ones(K) * 0.5, size=n)
Y64 = rs.poisson(B @ X * 0.6).astype(np.float64)
Y32 = Y64.astype(np.float32)
ref32 = set(orc.select_hvg(Y32, n_top).tolist())
ref64 = set(orc.select_hvg(Y64, n_top).tolist())
mean, var = g._gene_moments(Y32)
dev = set(g._hvg_from_moments(mean, var, n_top, 0.0125, 3.0, 0.5).tolist())
m32, v32 = mean.astype(np.float32), var.astype(np.float32)
dev32 = set(g._hvg_from_moments(m32, v32, n_top, 0.0125, 3.0, 0.5).tolist())
print(f"n {n} G {G} n_top {n_top}: reference f32 vs f64 differ in {len(ref32 ^ ref64) // 2} genes; device vs reference-f64 {len(dev ^ ref64) // 2}; "
      f"device vs reference-f32 {len(dev ^ ref32) // 2}; device moments rounded to float32 vs reference-f32 {len(dev32 ^ ref32) // 2}")
