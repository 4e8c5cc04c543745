"""productQuantizer.Fit timing (product.go:175-236): K = 256 centroids per sub-quantizer on the first 10 000 rows of the
C4 data (TriggerThreshold's maximum, models/quantizer.go:62), all M sub-quantizers in one set of launches."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vectorstore as vs

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--M", default="8,32,192")
ap.add_argument("--K", type=int, default=256)
ap.add_argument("--train", type=int, default=10000)
ap.add_argument("--dist", default="latent:24")
a = ap.parse_args()
train = bench.gen_rows(a.train, a.dim, 20250620, a.dist, "cuda:0").cpu().numpy()
out = {"dim": a.dim, "K": a.K, "rows": a.train, "fit_s": {}}
for M in [int(v) for v in a.M.split(",")]:
    best = None
    for rep in range(3):
        pq = vs.ProductQuantizer("cosine", vs.ProductQuantizerParameters(a.K, M, a.train), a.dim)
        x = train.copy()
        torch.cuda.synchronize()
        t0 = time.time()
        pq.Fit(x, np.arange(M) * 7 % a.train, alias=True)
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
        pq.close()
    out["fit_s"]["M=%d" % M] = round(best, 4)
print(json.dumps(out))
