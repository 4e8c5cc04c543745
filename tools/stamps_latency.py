"""Where a hop of a SMALL call spends its time (SDB_STAMPS diagnostic build, SEMADB_AMD_LIB=build/stamps/libsemadb_amd.so):
per-phase s_memtime sums of the walker wave, one wave per query and workgroup per query."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from semadb_amd import vamana

n, d = 1000000, 384
base = bench.gen_rows(n, d, 20250620, "latent:24", "cuda:0")
ix = vamana.NewIndexVamana("pv", vamana.IndexVectorVamanaParameters(d, "cosine", 75, 64, 1.2), capacity=n + 1)
ix.set_start(bench.start_vector(d))
ix.insert_batch(None, base)
queries = bench.gen_rows(4096, d, 20250621, "latent:24", "cuda:0")
for mode in (1, 2):
    ix.set_tuning("wide_walk", mode)
    for nq in (1, 64):
        acc = []
        hops = []
        for rep in range(8):
            q = queries[rep * nq:(rep + 1) * nq].contiguous()
            ids, dd, c, tr = ix.search_batch(q, 10, 75, trace=True, visit_cap=44)
            torch.cuda.synchronize()
            acc.append(tr.visit_ids.cpu().numpy().astype(np.float64))
            hops.append(tr.n_hop.cpu().numpy().astype(np.float64))
        full = np.concatenate(acc)
        nh = np.concatenate(hops).mean()
        v = full[:, :4]
        sub = full[:, 4:7]
        if mode == 2:
            w = full[:, 24:28].mean(axis=0) / nh
            mm = full[:, 8:12].mean(axis=0) / nh
            print("   walker, per hop: pick + guess %.0f, marker's verdict %.0f, late guess %.0f, naming %.0f | merge: preamble %.0f few %.0f per-point %.0f scatter %.0f" % (*w, *mm))
            print("   work ahead by wave, per hop:", " ".join("%d:%.0f" % (w_, full[:, 28 + w_].mean() / nh) for w_ in range(1, 16)))
            for name, o in (("marker", 12), ("computing wave", 18)):
                h = full[:, o:o + 5].mean(axis=0)
                packed = full[:, o + 5].astype(np.uint64)
                print("      inside the work ahead, per hop: adjacency row %.0f, rows %.0f" % ((packed >> np.uint64(32)).mean() / nh, (packed & np.uint64(0xFFFFFFFF)).mean() / nh))
                print("   %s, per hop: shares %.0f, waiting for the walker's word %.0f, work ahead %.0f (%.1f rows named per walk)"
                      % (name, h[1] / nh, h[2] / nh, h[3] / nh, h[4]))
        print("mode %d nq %4d hops %.1f | per hop (memtime ticks): adj %.0f atom %.0f vec %.0f ins %.0f total %.0f | inside vec: issue %.0f wait %.0f compute %.0f"
              % (mode, nq, nh, *(v.mean(axis=0) / nh), v.sum(axis=1).mean() / nh, *(sub.mean(axis=0) / nh)))
