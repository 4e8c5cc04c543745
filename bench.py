#!/usr/bin/env python3
"""bench.py -- QPS @ recall@10 >= 0.95, 1M x 384 Vamana search, batch = 1024 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--mode shards|replicas|c5] [--config c2|c3|c4]

One "step" = one batch of 1024 synthetic queries through the GPU-resident greedy search
(IndexVamana.Search semantics, searchSize 75, degreeBound 64, alpha 1.2, cosine, k = 10) with the
index, the queries and the results in HBM.  Index = BASELINE configs[1] (C2).

N > 1 (one rank per GPU, launched by torch.distributed.run), SURVEY 8e.  The default (--mode all) measures the three
modes below back to back and prints ONE line: value / ms_per_step / roofline are the primary mode's (shards, the
north-star mode), the other two sit under config.modes with the same fields; config.exchange names the transport
that carried the gather (RCCL inside libsemadb_amd.so, or the torch.distributed fallback) and config.ranks_seen the
ranks whose blocks the merge saw.
  --mode shards   (the north-star mode) the SAME 1M rows split into N contiguous shards of 1M/N,
                  one per GPU, each with its own graph; every shard answers every query; the per-shard
                  top-k blocks are exchanged with one RCCL all-gather issued by libsemadb_amd.so on its own
                  stream (sdb_cluster_search_batch) and merged with the reference's cluster rule
                  (cluster/actions.go:291-376).  value = merged, user-visible queries/s.  "strong".
  --mode replicas the full 1M index on every GPU, each batch split N ways, results gathered.  "strong".
  --mode c5       C5's shape: a shard of --c5-rows per GPU (12.5M: 100M over 8 GPUs; the database grows with N),
                  exchange as in `shards`.  value = merged queries/s.  "weak".

The JSON line also carries:
  roofline        algorithmic HBM bytes of the K2 kernel (n_dist*d*4 + n_edges*4 summed over the batch,
                  counted on device in a separate pass and identical to the oracle's counts) / its
                  HIP-event duration in the timed loop
  build_roofline  the same for the index build (C3): SURVEY 8d bytes / build seconds
  cpu_baseline    the reference algorithm (C restatement, AVX2 transcription of distance/asm/*.s,
                  oracle/) on this box's host cores over a bounded sample of the same batches --
                  a reported baseline, also used to check id parity at full size
  config.host_qps / config.batcher_qps   the rates with queries and results in host memory (SURVEY 8d's
                  "including H2D of queries and D2H of results"); never `value`

--config c3 reports the build (inserts/s) and --config c4 the product-quantized search (10M x 768) as the
value; both are secondary configurations, the driver runs the default.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s
LDS_B32_READS_PER_S = 75e12 / 4  # MI355X_MICROARCH.md: ~75 TB/s aggregate for ds_read_b32 with every CU streaming
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: packed FP32 FMA, 256 CUs x 256 flop/clk x 2.4 GHz


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def gen_rows(n, d, seed, dist, dev):
    """Synthetic float32 rows, L2-normalised (cosine needs it, docs/content/docs/concepts/distance.md:15).

    gaussian   i.i.d. N(0,1): the hard case -- at d = 384 no graph index reaches useful recall on it
               (measured: recall@10 = 0.04 at searchSize 75), so it cannot carry a recall-gated metric.
    latent:K   rows = z W + 0.1 eps with z in R^K: embedding-like data of intrinsic dimension K; the
               mixing matrix W (seed 1) is shared by base rows and queries.
    """
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if dist == "gaussian":
        x = torch.randn(n, d, generator=g, device=dev)
    else:
        k = int(dist.split(":")[1])
        gw = torch.Generator(device=dev)
        gw.manual_seed(1)
        w = torch.randn(k, d, generator=gw, device=dev)
        z = torch.randn(n, k, generator=g, device=dev)
        x = z @ w
        x += 0.1 * torch.randn(n, d, generator=g, device=dev)
    return torch.nn.functional.normalize(x, dim=1).contiguous()


def start_vector(d):
    g = torch.Generator().manual_seed(20250622)
    v = torch.rand(d, generator=g) * 2 - 1  # vamana.go:99-110
    return (v / v.norm()).numpy().astype(np.float32)


def exact_topk(queries, base, k, chunk=262144):
    """brute-force ground truth (cosine on unit rows = max dot); returns (scores, row indices)"""
    best_s, best_i = None, None
    for s in range(0, base.shape[0], chunk):
        sims = queries @ base[s:s + chunk].T
        ts, ti = sims.topk(min(k, sims.shape[1]), dim=1)
        ti = ti + s
        if best_s is None:
            best_s, best_i = ts, ti
        else:
            cs, ci = torch.cat([best_s, ts], 1), torch.cat([best_i, ti], 1)
            sel = cs.topk(k, dim=1).indices
            best_s, best_i = cs.gather(1, sel), ci.gather(1, sel)
    return best_s, best_i


def build_index(a, base, dev_index, name="bench", strict=True):
    from semadb_amd import vamana
    d = base.shape[1]
    params = vamana.IndexVectorVamanaParameters(d, a.metric, a.search_size, a.degree_bound, a.alpha)
    ix = vamana.NewIndexVamana(name, params, device=dev_index, capacity=base.shape[0] + 1, strict=strict)
    ix.set_start(start_vector(d))
    if os.environ.get("BENCH_NO_TILE"):  # measurement only (tools/pmc_build.sh): the one-wave prune of new nodes
        ix.set_tuning("no_tile", int(os.environ["BENCH_NO_TILE"]))
    for kv in filter(None, os.environ.get("BENCH_TUNE", "").split(",")):  # measurement only: "wide_hash=1,hash_limit=3000"
        key, value = kv.split("=")
        ix.set_tuning(key, int(value))
        log("tuning %s = %s (A/B measurement, not the shipped default)" % (key, value))
    torch.cuda.synchronize()
    t1 = time.time()
    ix.insert_batch(None, base)  # ids 2..n+1 ; K4 on device
    torch.cuda.synchronize()
    return ix, time.time() - t1


def build_roofline(ix, n, d, build_s):
    """bytes = sum over inserts of (search bytes + prune pair-distance rows * d * 4) / build time.
    Counters come from the device (sdb_index_build_stats); pair distances served from a cache are not billed."""
    st = ix.build_stats()
    search_b = st["search_n_dist"] * d * 4 + st["search_n_edges"] * 4
    pair_b = (st["prune_pairs"] + st["backedge_pairs"]) * d * 4
    alg = search_b + pair_b
    # what has to come from HBM at least once: the searches' rows, and each prune's candidate rows once
    unique_b = search_b + (st["staged_rows"] + st["backedge_pairs"]) * d * 4 if st["staged_rows"] else None
    ach = alg / build_s / 1e9
    # `frac` prices the bytes that have to come from HBM at least once (the searches' rows, each prune's candidate rows
    # once) against the build's wall clock; the all-pairs figure -- pair distances the prunes take from rows staged in
    # LDS are no HBM traffic -- stays beside it as `frac_mixed`
    uni = (unique_b if unique_b else alg) / build_s / 1e9
    out = {"bound": "hbm", "achieved": round(uni, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(uni / HBM_PEAK_GBS, 4), "achieved_mixed": round(ach, 1), "frac_mixed": round(ach / HBM_PEAK_GBS, 4),
           "traffic": None,
           "algorithmic_bytes": int(alg), "search_bytes": int(search_b), "prune_pair_bytes": int(pair_b),
           "build_s": round(build_s, 3), "inserts_per_s": round(n / build_s, 1),
           "note": "whole build (all kernels, %d rounds), wall clock; the prunes take pair distances from rows "
                   "staged in LDS, so the algorithmic rate is not an HBM rate for that part" % st["rounds"],
           "counters": st}
    if unique_b:
        out["hbm_unique_bytes"] = int(unique_b)
    return out


class Exchange:
    """The gather + merge step behind one interface: the product path (RCCL inside libsemadb_amd.so) or, for
    functional runs on a box with fewer GPUs than ranks, any torch.distributed backend (BENCH_BACKEND=gloo)."""

    def __init__(self, backend, dev, dev_index):
        import torch.distributed as dist
        from semadb_amd import cluster
        self.cluster, self.dist, self.dev, self.dev_index = cluster, dist, dev, dev_index
        self.world = dist.get_world_size()
        self.native = backend == "nccl"
        self.note = None
        if self.native:
            # every rank must end up on the same transport: agree on whether the library's communicator came up
            ok = 1
            try:
                self.cl = cluster.Cluster.from_torch_distributed(dev_index)
            except Exception as e:  # e.g. an RCCL that refuses the bootstrap: measure over torch's RCCL instead, and say so
                ok, self.note = 0, "sdb_cluster_create failed on rank %d: %r" % (dist.get_rank(), e)
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if ok:
                    self.cl.close()
                self.native = False
                self.note = self.note or "sdb_cluster_create failed on another rank"
                log("exchange falls back to torch.distributed all_gather_into_tensor:", self.note)
        self.seq = 0

    def search(self, ix, q, k, L):
        """ClusterNode.SearchPoints for this rank's shard -> merged (ids, dists, shards, counts) on device,
        valid after join()"""
        if self.native:
            return self.cl.search_batch(ix, q, k, L)
        # the torch.distributed transport: same block, same tags, same check + merge (blocking)
        per = self.cluster.shard_limit(k, self.world, 75)
        blk = self.cluster.PackedTopK(q.shape[0], per, self.dev)
        ix.search_batch(q, per, L, out=blk.out())
        self.seq += 1
        return blk.exchange(k, seq=self.seq, ticket=self.seq, queries=q)

    def join(self):
        if self.native:
            self.cl.synchronize()

    def close(self):
        if self.native:
            self.cl.close()


def rank_identity(dev_index, ex=None):
    """What this rank is, for the N > 1 line's self-diagnosis (config.rccl): its device ordinal and PCI bus id as the
    runtime reports them and -- when the library's communicator is up -- what sdb_cluster_info and
    sdb_cluster_transport say about it.  Gathered from every rank with all_gather_object."""
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
          "device_ordinal": dev_index, "pci_bus_id": None, "pid": os.getpid()}
    try:
        props = torch.cuda.get_device_properties(dev_index)
        if hasattr(props, "pci_bus_id"):
            me["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id,
                                                   getattr(props, "pci_device_id", 0))
        me["device_name"] = props.name
    except Exception:
        pass
    if ex is not None and getattr(ex, "native", False):
        try:
            from semadb_amd._lib import lib
            r, w, dv = C.c_int(-1), C.c_int(-1), C.c_int(-1)
            lib().sdb_cluster_info(ex.cl._h, C.byref(r), C.byref(w), C.byref(dv))
            me["cluster_rank"], me["cluster_world"], me["cluster_device"] = r.value, w.value, dv.value
            me["transport"] = ex.cl.transport()
        except Exception as e:
            me["cluster_error"] = repr(e)
    return me


def judge_ranks(infos, world, native, shared_device_ok=False):
    """-> (summary for config.rccl, reason the line is invalid or None).  Invalid: the ranks the communicator reports are
    not exactly 0 .. N-1, a rank's communicator disagrees on the world size, or two ranks drive the same device (unless
    the run is a functional one over gloo, where the ranks share a GPU on purpose)."""
    import re
    infos = sorted(infos, key=lambda r: r.get("rank", 0))
    out = {"ranks": [{k: r.get(k) for k in ("rank", "local_rank", "device_ordinal", "pci_bus_id", "cluster_rank",
                                            "cluster_world", "cluster_device")} for r in infos]}
    bad = None
    if [r.get("rank") for r in infos] != list(range(world)):
        bad = "ranks gathered %s, expected 0..%d" % ([r.get("rank") for r in infos], world - 1)
    if native:
        seen = sorted(r.get("cluster_rank", -1) for r in infos)
        out["ranks_seen_by_cluster_info"] = seen
        if seen != list(range(world)):
            bad = bad or "sdb_cluster_info reports ranks %s, expected 0..%d" % (seen, world - 1)
        if any(r.get("cluster_world") != world for r in infos):
            bad = bad or "a rank's communicator has %s ranks, expected %d" % ([r.get("cluster_world") for r in infos], world)
        t = next((r.get("transport") for r in infos if r.get("transport")), None)
        out["transport"] = t
        m = re.search(r"rccl (\d+\.\d+\.\d+)", t or "")
        out["nccl_version"] = m.group(1) if m else None
    devs = [(r.get("pci_bus_id") or "ordinal %s" % r.get("device_ordinal")) for r in infos]
    out["shared_device_ok"] = bool(shared_device_ok)
    if len(set(devs)) != len(devs) and not shared_device_ok:
        bad = bad or "two ranks drive the same device: %s" % devs
    return out, bad


def torchrun_command(gpus, argv, port=None):
    """The command line the driver itself uses for N > 1 (one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def relaunch_under_torchrun(gpus, argv):
    import subprocess
    cmd = torchrun_command(gpus, argv)
    log("no WORLD_SIZE in the environment and --gpus %d: starting the ranks:" % gpus, " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:            # rank 0 prints the one JSON line; anything else on stdout passes through too
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4"])
    ap.add_argument("--mode", default="all", choices=["all", "shards", "replicas", "c5"])
    ap.add_argument("--rows", type=int, default=None, help="rows in the database (--mode c5 alone: per GPU)")
    ap.add_argument("--c5-rows", type=int, default=12_500_000, help="rows per GPU of the c5 mode inside --mode all")
    ap.add_argument("--c4-rows", type=int, default=10_000_000,
                    help="rows of the quantized point in the default line: BASELINE configs[3] is 10M x 768 (0: skip)")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--search-size", type=int, default=75)
    ap.add_argument("--degree-bound", type=int, default=64)
    ap.add_argument("--alpha", type=float, default=1.2)
    ap.add_argument("--metric", default="cosine")
    ap.add_argument("--dist", default="latent:24")
    ap.add_argument("--recall-batches", type=int, default=10, help="query batches the recall phase walks")
    ap.add_argument("--timed-batches", type=int, default=20, help="distinct batches of the timed loop (not walked before)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the gaussian / latent:28 secondary points")
    ap.add_argument("--no-host-rates", action="store_true")
    ap.add_argument("--cpu-repeat", type=int, default=2)
    ap.add_argument("--pq-m", default="8,192", help="c4: sub-vector counts to measure")
    a = ap.parse_args()
    if a.rows is None:
        a.rows = 10_000_000 if a.config == "c4" else 1_000_000
    if a.dim is None:
        a.dim = 768 if a.config == "c4" else 384

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python3 bench.py --gpus N` without a launcher: start the ranks ourselves (the reference's counterpart simply
        # fans out, cluster/actions.go:316-351).  A CHILD process, started before anything here has touched the GPU
        # (never an exec from a process that has); its stdout -- rank 0's one JSON line -- is relayed, its rc returned.
        sys.exit(relaunch_under_torchrun(a.gpus, sys.argv[1:]))
    if world != a.gpus:
        a.gpus = world
    # BENCH_BACKEND=gloo + more ranks than GPUs is a functional test of the N > 1 path on a 1-GPU box
    # (ranks share the device; RCCL itself refuses two ranks on one GPU).  The driver never sets it.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index
    # BENCH_FORCE_EXCHANGE=1: take the N > 1 code path (RCCL all-gather on the exchange stream + merge) with a
    # single rank, so the real RCCL calls can be exercised on a 1-GPU box.  The driver never sets it.
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_EXCHANGE") == "1"
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if not use_dist:
            return
        if backend == "nccl":
            dist.barrier(device_ids=[dev_index])
        else:
            dist.barrier()

    ctx = dict(rank=rank, world=world, dev=dev, dev_index=dev_index, backend=backend, use_dist=use_dist, dist=dist,
               barrier=barrier)
    if a.config == "c3":
        result = run_c3(a, ctx)
    elif a.config == "c4":
        result = run_c4(a, ctx)
    else:
        result = run_c2(a, ctx)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        barrier()
        dist.destroy_process_group()
    if rank == 0 and result.get("invalid"):
        sys.exit(3)


# ------------------------------------------------------------------------------------------------------------
# C2: the headline
# ------------------------------------------------------------------------------------------------------------
def run_c2(a, ctx):
    """One line for the whole SURVEY 8e record: N = 1 -> the headline; N > 1 and --mode all -> shards (primary: value,
    ms_per_step, roofline), then replicas and c5 under config.modes."""
    world = ctx["world"]
    ctx["ex"] = Exchange(ctx["backend"], ctx["dev"], ctx["dev_index"]) if ctx["use_dist"] else None
    # BENCH_REHEARSE_ALL=1: ONE rank walks the whole N > 1 schedule (shards, replicas, c5 at --c5-rows per GPU; none of
    # the single-GPU extras), so that the wall time of the driver's 8-GPU lease can be budgeted on a 1-GPU box -- per-rank
    # work does not depend on N except for the shards mode's 1/N split, which makes N = 1 the slowest case.  With
    # BENCH_FORCE_EXCHANGE=1 on top the exchange is the library's RCCL all-gather (one rank).  The driver sets neither.
    ctx["rehearse"] = os.environ.get("BENCH_REHEARSE_ALL") == "1"
    t_all = time.time()
    try:
        if (world == 1 and not ctx["rehearse"]) or a.mode != "all":
            res = measure_mode(a, ctx, "shards" if (world == 1 or a.mode == "all") else a.mode, a.rows, primary=True)
            res["config"]["wall_s_total"] = round(time.time() - t_all, 1)
            return res
        res = measure_mode(a, ctx, "shards", a.rows, primary=True)
        walls = ["shards: " + res["config"]["wall_s_by_phase"]]
        modes = {}
        for m, rows in (("replicas", a.rows), ("c5", a.c5_rows)):
            r = measure_mode(a, ctx, m, rows, primary=False)
            cfg = r["config"]
            walls.append("%s: %s" % (m, cfg["wall_s_by_phase"]))
            res["config"]["%s_qps" % m] = r["value"]  # scalars: nested objects do not survive the driver's record
            res["config"]["%s_recall_at_10" % m] = cfg["recall_at_10"]
            res["config"]["%s_roofline_frac" % m] = r["roofline"]["frac"]
            modes[m] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "scaling": r["scaling"],
                        "workload": cfg["workload"], "parallelism": cfg["parallelism"], "dataset": cfg["dataset"],
                        "recall_at_10": cfg["recall_at_10"], "build_s": cfg["build_s"], "mean_n_dist": cfg["mean_n_dist"],
                        "exchange": cfg.get("exchange"), "exchange_transport": cfg.get("exchange_transport"),
                        "ranks_seen": cfg.get("ranks_seen"),
                        "per_shard_walk_qps": cfg.get("per_shard_walk_qps"), "roofline": r["roofline"]}
            if r.get("invalid"):
                modes[m]["invalid"] = r["invalid"]
        res["config"]["primary_mode"] = "shards"
        res["config"]["wall_s_by_phase"] = "; ".join(walls)
        res["config"]["wall_s_total"] = round(time.time() - t_all, 1)
        if ctx["rehearse"]:
            res["config"]["rehearsal"] = "BENCH_REHEARSE_ALL=1: one rank walked the N > 1 schedule; not a measurement of the metric"
        res["config"]["modes"] = modes
        return res
    finally:
        if ctx["ex"] is not None:
            ctx["ex"].close()


def measure_mode(a, ctx, mode, rows, primary):
    from semadb_amd import flat
    rank, world, dev, dev_index = ctx["rank"], ctx["world"], ctx["dev"], ctx["dev_index"]
    use_dist, dist, barrier = ctx["use_dist"], ctx["dist"], ctx["barrier"]
    d, nq, k, L = a.dim, a.batch, a.k, a.search_size
    a_rows = rows

    phases = []  # (name, seconds) of this mode's wall time, in order
    t_phase = [time.time()]

    def phase(name):
        torch.cuda.synchronize()
        now = time.time()
        phases.append("%s %.1f" % (name, now - t_phase[0]))
        t_phase[0] = now

    # ---- data.  shards / replicas: ONE database of --rows rows (seed 20250620), identical on every rank;
    # c5: shard `rank` = --rows rows of its own (seed 20250620 + rank, SURVEY 8d per-shard offset)
    t0 = time.time()
    if mode == "c5":
        base = gen_rows(a_rows, d, 20250620 + rank, a.dist, dev)
        n_total = a_rows * world
    else:
        full = gen_rows(a_rows, d, 20250620, a.dist, dev)
        n_total = a_rows
        if mode == "shards" and world > 1:  # contiguous ranges, like the reference fills shards in order
            lo, hi = rank * a_rows // world, (rank + 1) * a_rows // world
            base = full[lo:hi].contiguous()
            del full
        else:
            base = full
    n = base.shape[0]
    nb_recall, nb_timed = a.recall_batches, a.timed_batches
    queries = gen_rows((nb_recall + nb_timed) * nq, d, 20250621, a.dist, dev).view(nb_recall + nb_timed, nq, d)
    t1 = time.time()
    phase("data")
    ix, build_s = build_index(a, base, dev_index)
    phase("build")
    n_nodes, n_edges, _ = ix.stats()
    broof = build_roofline(ix, n, d, build_s)
    log("rank %d: data %.1fs, build %.2fs (%.0f inserts/s), avg degree %.2f" %
        (rank, t1 - t0, build_s, n / build_s, n_edges / n_nodes))

    ex = ctx["ex"]
    split = mode == "replicas" and world > 1
    q_lo, q_hi = (rank * nq // world, (rank + 1) * nq // world) if split else (0, nq)

    def step(b):
        """one batch through the hot path; returns (ids, dists, shards, counts) as enqueued device tensors"""
        q = queries[b]
        if split:  # replicas: this rank's slice of the batch on the full index, then a plain gather
            ids, dists, counts, _ = ix.search_batch(q[q_lo:q_hi], k, L)
            return ids, dists, None, counts
        if ex is None:
            ids, dists, counts, _ = ix.search_batch(q, k, L)
            return ids, dists, None, counts
        return ex.search(ix, q, k, L)

    def gather_replicas(ids, dists, counts):
        """replicas: every rank receives the answers to the whole batch (padded slices, rank-major)"""
        outs = []
        for t in (ids, dists, counts):
            g = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(g, t.contiguous())
            outs.append(torch.cat(g, 0))
        return outs

    # ---- recall@10 against exact ground truth over the whole database, on the recall batches
    hits = total = 0
    ranks_seen = set()
    for b in range(nb_recall):
        m_ids, m_d, m_sh, m_c = step(b)
        if ex is not None and not split:
            ex.join()
        torch.cuda.synchronize()
        if split:
            m_ids, m_d, m_c = gather_replicas(m_ids, m_d, m_c)
            ts, ti = exact_topk(queries[b], base, k)
            truth, got = ti + 2, m_ids.to(torch.int64)
        elif use_dist:
            ts, ti = exact_topk(queries[b], base, k)
            all_s = [torch.empty_like(ts) for _ in range(world)]
            all_i = [torch.empty_like(ti) for _ in range(world)]
            dist.all_gather(all_s, ts)
            dist.all_gather(all_i, ti)
            cs = torch.cat(all_s, 1)
            ci = torch.cat([(x + 2) + (r << 40) for r, x in enumerate(all_i)], 1)  # (shard, id) packed
            sel = cs.topk(k, dim=1).indices
            truth = ci.gather(1, sel)
            got = m_ids.to(torch.int64) + (m_sh.to(torch.int64) << 40)
            ranks_seen.update(int(v) for v in torch.unique(m_sh).tolist())
        else:
            truth = exact_topk(queries[b], base, k)[1] + 2
            got = m_ids.to(torch.int64)
        eq = (got.unsqueeze(2) == truth.unsqueeze(1)).any(2)
        hits += int(eq.sum().item())
        total += nq * k
    recall = hits / total
    log("recall@%d = %.4f at searchSize %d (%d queries)" % (k, recall, L, total // k))
    # cross-check of the ground truth itself: the device flat scan (IndexFlat.Search, flat.go:76-132, same
    # bit-exact distances) against the torch matmul top-k, first batch, this rank's shard
    f_ids, _, _ = flat.flat_search_batch(ix._h, d, queries[0], k, device=dev_index)
    t_ids = exact_topk(queries[0], base, k)[1] + 2
    truth_agree = float((f_ids.to(torch.int64).unsqueeze(2) == t_ids.unsqueeze(1)).any(2).float().mean().item())
    log("flat scan vs matmul ground truth agreement: %.4f" % truth_agree)
    # the exact scan as a measured point of its own (IndexFlat.Search of one query batch over this rank's rows)
    torch.cuda.synchronize()
    t_f = time.perf_counter()
    for _ in range(3):
        flat.flat_search_batch(ix._h, d, queries[0], k, device=dev_index)
    torch.cuda.synchronize()
    flat_ms = (time.perf_counter() - t_f) / 3 * 1e3
    log("flat exact scan: %.2f ms per %d queries over %d rows" % (flat_ms, nq, n))

    phase("recall+scan")
    # ---- timed region: batches the recall phase never walked, trace counters OFF, one batch at a time
    tb = [nb_recall + (i % nb_timed) for i in range(a.warmup + a.steps)]
    ix.set_profiling(True)
    for w in range(a.warmup):
        step(tb[w])
    if ex is not None:
        ex.join()
    torch.cuda.synchronize()
    barrier()
    ix.profile_read()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    last = None
    for s in range(a.steps):
        last = step(tb[a.warmup + s])
        if split:
            last = gather_replicas(last[0], last[1], last[3])
    if ex is not None:
        ex.join()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = ix.profile_read()[-a.steps:]
    ix.set_profiling(False)

    # ---- counter pass (separate from the timed loop): algorithmic bytes of every timed batch
    per_batch = {}
    nd_all, nh_all, nd_tail = [], [], []
    for b in sorted(set(tb[a.warmup:])):
        qq = queries[b][q_lo:q_hi]
        _, _, _, tr = ix.search_batch(qq, k, L, trace=True)
        nd = int(tr.n_dist.to(torch.int64).sum().item())
        ne = int(tr.n_edges.to(torch.int64).sum().item())
        per_batch[b] = nd * d * 4 + ne * 4
        nd_all.append(nd / qq.shape[0])
        nh_all.append(float(tr.n_hop.float().mean().item()))
        # how uneven the walks of ONE batch are: a batch ends on its longest walk
        ndq = tr.n_dist.float()
        nd_tail.append(float(ndq.max().item() / ndq.mean().item()))
    alg_bytes = [per_batch[b] for b in tb[a.warmup:]][-len(kernel_ms):]
    achieved = float(np.sum(alg_bytes) / (np.sum(kernel_ms) * 1e-3) / 1e9) if len(kernel_ms) else 0.0

    phase("timed+counters")
    # value: user-visible queries answered per second -- every query counts once however many shards walked it
    qps = nq * a.steps / elapsed
    scaling = "weak" if mode == "c5" else "strong"
    if world == 1:
        par = "1 gpu"
        scaling = "weak"
    elif mode == "replicas":
        par = "replicas x%d: full index per GPU, batch split %d ways, all-gather of the answers" % (world, world)
    else:
        par = "shard-per-gpu x%d, every shard answers every query, one RCCL all-gather per batch issued by " \
              "libsemadb_amd.so (sdb_cluster_search_batch) + device top-k merge" % world
    result = {
        "metric": "QPS @ recall@10>=0.95, 1Mx384 Vamana search, batch=1024",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "vectorVamana search %dx%d %s, searchSize=%d degreeBound=%d alpha=%.1f, batch=%d, k=%d, "
                        "%d shard(s) x %d rows (%d rows in all)" % (n_total, d, a.metric, L, a.degree_bound, a.alpha,
                                                                   nq, k, 1 if mode == "replicas" else world, n, n_total),
            "mode": mode,
            "dataset": "%s seed 20250620%s, queries seed 20250621 (not in base set)" %
                       (a.dist, "(+rank)" if mode == "c5" else ""),
            "recall_at_10": round(recall, 4),
            "recall_gate": 0.95,
            "ground_truth": "exact brute force over the whole database (torch matmul top-k); agreement with the "
                            "device flat scan %.4f" % truth_agree,
            "flat_scan_ms": round(flat_ms, 2),
            "flat_scan": "IndexFlat.Search (exact scan, same distance bits) of one %d-query batch over %d rows: "
                         "%.1f G pairs/s, %.1f useful TFLOP/s" % (nq, n, nq * n / flat_ms / 1e6, 2.0 * nq * n * d / flat_ms / 1e9),
            "search_size": L,
            "parallelism": par,
            "value_definition": "user-visible queries answered per second (each query counted once, after the "
                                "shard merge); index, queries and results resident in HBM",
            "timed_batches": "%d distinct batches never walked before the timed loop; trace counters off" % nb_timed,
            "per_shard_walk_qps": round(world * nq * a.steps / elapsed, 1) if (world > 1 and not split) else None,
            "per_shard_limit": (min(k, 75, int(k * (1.0 / world) * 1.42 + 10)) if not split else k),  # actions.go:291-299
            "build_s": round(build_s, 2),
            "build_inserts_per_s": round(n / build_s, 1),
            "avg_degree": round(n_edges / n_nodes, 2),
            "mean_n_dist": round(float(np.mean(nd_all)), 1),
            "mean_n_hop": round(float(np.mean(nh_all)), 1),
            # the longest walk of a batch over its mean walk, in distance evaluations: what one batch in flight idles for at
            # its end (two batches in flight hide it: two_batches_in_flight_qps)
            "longest_walk_over_mean": round(float(np.mean(nd_tail)), 3),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_greedy_search",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "traffic_measured_in_run": False,
            "algorithmic_bytes_per_launch": int(np.mean(alg_bytes)) if alg_bytes else 0,
            "kernel_ms_avg": round(float(np.mean(kernel_ms)), 4) if len(kernel_ms) else None,
            "caveat": "the algorithmic rate counts every row the walk reads; about 17 % of a 1.5 GB slab plus the hub "
                      "rows are Infinity-Cache hits (MI355X_MICROARCH.md: ~6.3 TB/s achievable DRAM stream), so the "
                      "DRAM-side rate is ~6.0 TB/s; a dependency-free gather of the same rows reaches 7.0 TB/s on the "
                      "same box (profiles/*gather_ceiling.json)",
        },
        "build_roofline": broof,
    }
    if split:
        result["config"]["exchange"] = "torch.distributed all_gather of the answers (%s); no merge" % ctx["backend"]
        result["config"]["ranks_seen"] = list(range(world))
    if use_dist:
        # the N > 1 line diagnoses itself (the first 8-GPU run will be the driver's, unattended): who the ranks are, what
        # the communicator says, and what the exchange alone costs
        infos = [None] * world
        dist.all_gather_object(infos, rank_identity(dev_index, ex))
        rccl, bad = judge_ranks(infos, world, bool(ex is not None and ex.native),
                                shared_device_ok=(ctx["backend"] != "nccl" or os.environ.get("BENCH_FORCE_EXCHANGE") == "1"))
        if ex is not None and not split:
            try:  # the exchange alone: all-gather + merge of a block that is already there, per batch
                from semadb_amd import cluster as _cl
                per = _cl.shard_limit(k, world, 75)
                blk = _cl.PackedTopK(nq, per, dev)
                reps_x = 20
                for i in range(reps_x + 3):
                    if i == 3:
                        torch.cuda.synchronize()
                        barrier()
                        t_x = time.perf_counter()
                    if ex.native:
                        ex.cl.allgather_merge(blk, k)
                        ex.cl.synchronize()
                    else:
                        ex.seq += 1
                        blk.exchange(k, seq=ex.seq, ticket=ex.seq, queries=None)
                torch.cuda.synchronize()
                rccl["allgather_merge_us_per_batch"] = round((time.perf_counter() - t_x) / reps_x * 1e6, 1)
                rccl["allgather_bytes_per_rank"] = int(blk.buf.numel())
            except Exception as e:
                rccl["allgather_merge_error"] = repr(e)
        result["config"]["rccl"] = rccl
        if bad:
            result["invalid"] = bad
    if ex is not None and not split:
        result["config"]["ranks_seen"] = sorted(ranks_seen)
        if ex.native:  # the library's own words: RCCL version, the librccl this process loaded, communicator size, rank
            result["config"]["exchange_transport"] = ex.cl.transport()
        result["config"]["exchange"] = ("libsemadb_amd.so: sdb_cluster_search_batch (ncclAllGather on the library's stream)"
                                        if ex.native else "torch.distributed all_gather_into_tensor (%s) + sdb_cluster_merge_gathered (tag check + merge)%s" %
                                        (ctx["backend"], "; " + ex.note if ex.note else ""))
    if ex is not None and not split and sorted(ranks_seen) != list(range(world)):
        result["invalid"] = "the merged answers name shards %s, expected every one of 0..%d" % (sorted(ranks_seen), world - 1)
    if recall < 0.95:  # the metric is recall-gated: a line below the gate is not a measurement of it
        result["invalid"] = "recall@10 %.4f is below the metric's 0.95 gate" % recall
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            if rec.get("workload_n") == n and rec.get("dim") == d and rec.get("dist") == a.dist and world == 1:
                result["roofline"]["traffic"] = rec["hbm_bytes_per_launch"]
                result["roofline"]["traffic_source"] = "profile-derived, not measured in this run: " + str(rec.get("source"))
        except Exception:
            pass

    # the same protocol on the graph the REFERENCE's sequential insert schedule builds from the same rows (415 s on the
    # device: not part of this run) -- tools/reference_schedule_graph.py wrote the record; it is repeated here, labelled
    ref = os.path.join(ROOT, "profiles", "r06_refsched_1m.json")
    if os.path.exists(ref) and world == 1 and primary:
        try:
            rec = json.load(open(ref))
            if rec.get("rows_reached") == n and a.dist in rec.get("workload", "") and "%d x %d" % (n, d) in rec["workload"]:
                g = rec["reference_schedule_graph"]
                result["config"]["reference_schedule_graph"] = {
                    "source": "profile-derived, not measured in this run: profiles/r06_refsched_1m.json "
                              "(tools/reference_schedule_graph.py; insert.go:16-68 one point after another, the first "
                              "%d rows equal to the oracle's graph edge for edge)" % rec["prefix"]["rows"],
                    "qps": g["qps"], "recall_at_10": g["recall_at_10"], "kernel_ms_avg": g["kernel_ms_avg"],
                    "frac_of_hbm_peak": g["frac_of_hbm_peak"], "mean_n_dist": g["mean_n_dist"],
                    "sequential_build_s": rec["sequential_build_s"],
                    "batched_graph_same_process": {k: rec["batched_graph"][k] for k in ("qps", "recall_at_10", "frac_of_hbm_peak", "mean_n_dist")}}
        except Exception:
            pass

    if rank == 0 and world == 1 and primary and not ctx.get("rehearse"):
        cfg = result["config"]
        # the timed loop once more, the same way (a run checks itself: the driver's N = 1 figures of its BENCH and SCALE
        # passes should agree like these two do)
        try:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for s_ in range(a.steps):
                step(tb[a.warmup + s_])
            torch.cuda.synchronize()
            v2 = nq * a.steps / (time.perf_counter() - t1)
            cfg["value_second_pass"] = round(v2, 1)
            cfg["value_passes_agree_within_3pct"] = bool(abs(v2 - result["value"]) <= 0.03 * result["value"])
        except Exception as e:
            cfg["value_second_pass"] = repr(e)
        # not the metric (one batch at a time): the same batches with two of them in flight on two streams, the way a
        # serving process (the host batcher) runs -- a second batch fills the SIMDs the first one's finished walks left
        try:
            streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            for st in streams:
                st.wait_stream(torch.cuda.current_stream())
            reps = max(a.steps, 20)
            for i in range(4):
                with torch.cuda.stream(streams[i % 2]):
                    step(tb[i % len(tb)])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for i in range(reps):
                with torch.cuda.stream(streams[i % 2]):
                    step(tb[a.warmup + i % a.steps])
            torch.cuda.synchronize()
            cfg["two_batches_in_flight_qps"] = round(nq * reps / (time.perf_counter() - t2), 1)
        except Exception as e:
            cfg["two_batches_in_flight_qps"] = repr(e)
        if not a.no_host_rates:
            try:
                cfg.update(host_rates(a, ix, queries[nb_recall:], k, L, last))
                # SURVEY 8d's protocol figure ("including H2D of queries and D2H of results") beside the headline
                result["value_host"] = cfg["host_qps"]
                result["value_host_definition"] = cfg["host_qps_definition"]
            except Exception as e:
                cfg["host_rates_error"] = repr(e)
        if not a.no_secondary:
            try:
                cfg["side_kernels"] = side_points(a, ix, base, queries, dev)
            except Exception as e:
                cfg["side_kernels"] = {"error": repr(e)}
            try:
                cfg["latency_ms"] = latency_points(a, ix, queries)
            except Exception as e:
                cfg["latency_ms"] = {"error": repr(e)}
        if not a.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(a, ix, queries[:nb_recall], k, L)
            except Exception as e:  # never lose the GPU line to a host-side problem
                result["cpu_baseline"] = {"error": repr(e)}
        # not the metric's default path: the two-precision hop (SDB_TUNE_SKETCH) on the same batches, checked against it.
        # (Last on this index: its streams and its copy of the rows must not disturb the figures above.)
        try:
            cfg["two_precision_hop"] = two_precision_point(a, ix, queries, tb, k, L, d, nq, result)
            tp = cfg["two_precision_hop"]
            if tp.get("identical_to_the_default_walk") and not tp.get("audit_contradicted") and "invalid" not in tp:
                result["value_two_precision"] = tp["qps"]
                result["value_two_precision_definition"] = (
                    "the same timed loop with SDB_TUNE_SKETCH = 1 (off by default; a float16 copy of the rows read first, "
                    "float32 rows only for neighbours AddWithLimit may keep): every answer of every timed batch compared "
                    "with the default walk's, bit for bit -- `value` is the default walk")
        except Exception as e:
            cfg["two_precision_hop"] = {"error": repr(e)}
    ix.close()
    del base
    torch.cuda.empty_cache()
    phase("extras")
    if rank == 0 and world == 1 and primary and not a.no_secondary and not ctx.get("rehearse"):
        try:
            result["config"]["secondary_datasets"] = secondary_points(a, dev, dev_index)
        except Exception as e:
            result["config"]["secondary_datasets"] = {"error": repr(e)}
        phase("secondary")
        if a.c4_rows:
            try:
                result["config"]["c4"] = c4_point(a, dev, dev_index)
            except Exception as e:
                result["config"]["c4"] = {"error": repr(e)}
            phase("c4")
    result["config"]["wall_s_by_phase"] = ", ".join(phases)
    flatten_summary(result)
    return result


def two_precision_point(a, ix, queries, tb, k, L, d, nq, result):
    """The timed loop again with SDB_TUNE_SKETCH = 1 (include/semadb_amd.h): the index keeps a float16 copy of its rows
    and a hop reads a new neighbour's float32 row only when the float16 distance does not prove that AddWithLimit
    discards it (distset.go:184).  Every answer of every timed batch is compared with the default walk's, bit for bit;
    a separate pass in audit mode evaluates every discarded neighbour exactly as well and counts contradictions."""
    distinct = sorted(set(tb[a.warmup:]))
    ref = {}
    for b in distinct:
        ids, dd, cnt, tr = ix.search_batch(queries[b], k, L, trace=True)
        ref[b] = (ids.clone(), dd.clone().view(torch.int32), cnt.clone(), tr.n_dist.clone(), tr.n_hop.clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.set_tuning("sketch", 1)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    out = {"knob": "SDB_TUNE_SKETCH = 1 (off by default): float16 copy of the rows, + 50 % of their memory, kept current by every commit",
           "copy_build_s": round(build_s, 4), "in_use": ix.sketch_stats()[2]}
    try:
        for b in range(2):
            ix.search_batch(queries[b], k, L)
        torch.cuda.synchronize()
        ix.set_profiling(True)
        ix.profile_read()
        t1 = time.perf_counter()
        for s_ in range(a.steps):
            ix.search_batch(queries[tb[a.warmup + s_]], k, L)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        kms = [float(v) for v in ix.profile_read()][-a.steps:]
        ix.set_profiling(False)
        # ... and with two batches in flight, like two_batches_in_flight_qps does for the default walk
        dev = queries.device
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        reps = max(a.steps, 20)
        for i in range(4):
            with torch.cuda.stream(streams[i % 2]):
                ix.search_batch(queries[tb[i % len(tb)]], k, L)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(reps):
            with torch.cuda.stream(streams[i % 2]):
                ix.search_batch(queries[tb[a.warmup + i % a.steps]], k, L)
        torch.cuda.synchronize()
        out["two_batches_in_flight_qps"] = round(nq * reps / (time.perf_counter() - t2), 1)
        same = True
        n_dist = 0
        for b in distinct:
            ids, dd, cnt, tr = ix.search_batch(queries[b], k, L, trace=True)
            r = ref[b]
            same &= bool(torch.equal(ids, r[0]) and torch.equal(dd.view(torch.int32), r[1]) and torch.equal(cnt, r[2]) and
                         torch.equal(tr.n_dist, r[3]) and torch.equal(tr.n_hop, r[4]))
            n_dist += int(tr.n_dist.to(torch.int64).sum().item())
        discarded_all = ix.sketch_stats()[0]
        ix.set_tuning("sketch", 2)  # audit: counters start at zero
        for b in distinct:
            ix.search_batch(queries[b], k, L)
        discarded, contradicted, _ = ix.sketch_stats()
        qps = nq * a.steps / dt
        per_launch_nd = n_dist / len(distinct)
        bytes_read = per_launch_nd * d * 2 + (per_launch_nd - discarded / len(distinct)) * d * 4  # (float16 rows: an upper bound)
        kavg = float(np.mean(kms)) if kms else 0.0
        out.update({"qps": round(qps, 1), "ms_per_step": round(dt / a.steps * 1e3, 4), "kernel_ms_avg": round(kavg, 4),
                    "speedup_over_value": round(qps / result["value"], 3),
                    "identical_to_the_default_walk": same,
                    "compared": "%d distinct timed batches: ids, distance bits, counts, n_dist, n_hop" % len(distinct),
                    "discarded_frac_of_evaluated": round(discarded / max(1, n_dist), 4),
                    "audit_contradicted": int(contradicted),
                    "row_bytes_read_per_launch_upper_bound": int(bytes_read),
                    "roofline": {"bound": "hbm", "kernel": "k_greedy_search<PlainDist<..., SK>>",
                                 "achieved": round(bytes_read / (kavg * 1e-3) / 1e9, 1) if kavg else None, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": round(bytes_read / (kavg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kavg else None,
                                 "note": "bytes actually read (float16 row of every evaluated neighbour + float32 row of "
                                         "the kept ones), not SURVEY 8d's n_dist x d x 4: the walk is no longer bound by HBM"}})
        if not same or contradicted:
            out["invalid"] = "the two-precision hop disagrees with the default walk"
        _ = discarded_all
    finally:
        ix.set_profiling(False)
        ix.set_tuning("sketch", 0)
    return out


def flatten_summary(result):
    """The driver's record keeps the scalars of `config` and drops nested objects: the figures a reader of that record
    needs are repeated as scalars (the nested objects stay for whoever reads the full line)."""
    cfg = result["config"]
    flat = {}
    if "value_host" in result:
        flat["value_host"] = result["value_host"]  # SURVEY 8d's protocol figure (H2D + D2H inside the timed region)
    tp = cfg.get("two_precision_hop")
    if isinstance(tp, dict) and "qps" in tp:
        flat["two_precision_hop_qps"] = tp["qps"]
        flat["two_precision_hop_identical_to_the_default_walk"] = tp["identical_to_the_default_walk"]
        flat["two_precision_hop_kernel_ms"] = tp["kernel_ms_avg"]
    rs = cfg.get("reference_schedule_graph")
    if isinstance(rs, dict):  # profile-derived (labelled in the nested object)
        flat["reference_schedule_graph_qps_from_profile"] = rs.get("qps")
        flat["reference_schedule_graph_recall_at_10_from_profile"] = rs.get("recall_at_10")
    sec = cfg.get("secondary_datasets") or {}
    for name, rec in sec.items():
        if isinstance(rec, dict) and "qps" in rec:
            key = name.replace(":", "")
            flat["%s_qps" % key] = rec["qps"]
            flat["%s_recall_at_10" % key] = rec["recall_at_10"]
            if "search_size_for_recall_0.95" in rec:
                flat["%s_search_size_for_recall_0.95" % key] = rec["search_size_for_recall_0.95"]
            if isinstance(rec.get("roofline"), dict):  # three datasets, three fractions (the headline's is roofline.frac)
                flat["%s_roofline_frac" % key] = rec["roofline"].get("frac")
                flat["%s_mean_n_dist" % key] = rec.get("mean_n_dist")
                flat["%s_kernel_ms" % key] = rec.get("kernel_ms_avg")
                flat["%s_algorithmic_bytes_per_launch" % key] = rec.get("algorithmic_bytes_per_launch")
                flat["%s_longest_walk_over_mean" % key] = rec.get("longest_walk_over_mean")
                flat["%s_two_batches_in_flight_frac" % key] = rec.get("two_batches_in_flight_frac")
                flat["%s_two_precision_hop_qps" % key] = rec.get("two_precision_hop_qps")
    c4tp = (cfg.get("c4") or {}).get("full_precision_two_precision_hop") if isinstance(cfg.get("c4"), dict) else None
    if isinstance(c4tp, dict) and "call_qps" in c4tp:
        flat["c4_full_precision_two_precision_hop_qps"] = c4tp["call_qps"]
        flat["c4_full_precision_two_precision_hop_recall_at_10"] = c4tp["recall_at_10"]
    hb = cfg.get("host_blocking_variants") or {}
    for name in ("staged", "pageable"):
        if isinstance(hb.get(name), dict):
            flat["host_qps_%s" % name] = hb[name].get("qps")
    c4 = cfg.get("c4") or {}
    for mk, rec in c4.items():
        if mk.startswith("M=") and isinstance(rec, dict):
            tag = "c4_" + mk.replace("=", "")
            for src, dst in (("call_qps", "call_qps"), ("kernel_ms", "kernel_ms"), ("recall_at_10", "recall_at_10"),
                             ("traffic_over_algorithmic", "traffic_over_algorithmic"), ("lookups_per_s", "lookups_per_s"),
                             ("bound_by", "bound_by")):
                if rec.get(src) is not None:
                    flat["%s_%s" % (tag, dst)] = rec[src]
            if isinstance(rec.get("roofline"), dict):
                flat["%s_roofline_frac" % tag] = rec["roofline"].get("frac")
                flat["%s_roofline_bound" % tag] = rec["roofline"].get("bound")
            if isinstance(rec.get("latency_roof"), dict):
                flat["%s_latency_roof_frac" % tag] = rec["latency_roof"].get("frac")
                flat["%s_dependent_fetch_ns" % tag] = (rec["latency_roof"].get("dependent_fetch_ns") or {}).get("loaded")
            if isinstance(rec.get("encode_kernel_roofline"), dict) and "frac" in rec["encode_kernel_roofline"]:
                flat["%s_encode_kernel_frac_of_fp32_vector_peak" % tag] = rec["encode_kernel_roofline"]["frac"]
            if rec.get("code_row_layout_bytes") is not None:
                flat["%s_code_row_layout_bytes" % tag] = rec["code_row_layout_bytes"]
    if isinstance(c4.get("full_precision"), dict):
        flat["c4_full_precision_call_qps"] = c4["full_precision"].get("call_qps")
        flat["c4_full_precision_recall_at_10"] = c4["full_precision"].get("recall_at_10")
    lat = cfg.get("latency_ms") or {}
    for key in ("1", "256"):
        v = (lat.get("workgroup_per_query") or {}).get(key) if isinstance(lat, dict) else None
        if isinstance(v, (int, float)):
            flat["latency_ms_%s_queries" % key] = v
        h = (lat.get("host_memory_call") or {}).get(key) if isinstance(lat, dict) else None
        if isinstance(h, (int, float)):
            flat["latency_ms_%s_queries_host_memory" % key] = h
    br = result.get("build_roofline")
    if isinstance(br, dict):
        for key in ("frac", "frac_mixed", "hbm_unique_bytes", "build_s"):
            if key in br:
                flat["build_roofline_%s" % key] = br[key]
    # scalars first: rebuild config with the summary in front
    result["config"] = {**{k: v for k, v in cfg.items() if k in ("workload", "mode", "dataset", "recall_at_10")}, **flat,
                        **{k: v for k, v in cfg.items() if k not in ("workload", "mode", "dataset", "recall_at_10")}}


def side_points(a, ix, base, queries, dev):
    """The path's other kernels on the headline's data, so that they are in the driver's record and not only in
    builder-run logs: K1 (sdb_distance_batch: 64 queries x all rows, row reuse on chip), the euclidean exact scan
    (IndexFlat.Search, packed-FMA kernel; the cosine scan is `flat_scan_ms` above), the filtered walk
    (search.go:33-51,93-95) with 10 / 1 000 filter ids per query (kernel time from HIP events inside the library; the
    whole call adds the upload of the filter ids and their translation to slots, on the device for a table with
    consecutive ids)."""
    from semadb_amd import distance, flat
    n, d = base.shape
    out = {}
    q64 = queries[0][:64].contiguous()
    for metric in ("cosine", "euclidean"):
        distance.distance_batch(metric, q64, base)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            r = distance.distance_batch(metric, q64, base)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        del r
        out["k1_64x%dx%d_%s" % (n, d, metric)] = {"ms": round(dt * 1e3, 3), "G_pairs_per_s": round(64 * n / dt / 1e9, 1)}
    fx = flat.NewIndexFlat(flat.IndexVectorFlatParameters(d, "euclidean"), capacity=n + 1)
    try:
        fx.set_vectors(None, base)
        qb = queries[0].contiguous()
        for _ in range(2):
            flat.flat_search_batch(fx._h, d, qb, a.k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            flat.flat_search_batch(fx._h, d, qb, a.k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        out["exact_scan_euclidean_%dx%dx%d" % (qb.shape[0], n, d)] = {
            "ms": round(dt * 1e3, 2), "G_pairs_per_s": round(qb.shape[0] * n / dt / 1e9, 1)}
    finally:
        fx.close()
    rng = np.random.default_rng(3)
    nq = queries.shape[1]
    ix.set_profiling(True)
    try:
        for size in (10, 1000):
            filt = [np.sort(rng.choice(n, size=size, replace=False).astype(np.uint64) + 2) for _ in range(nq)]
            off = np.zeros(nq + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(f) for f in filt])
            ids = np.concatenate(filt)
            for _ in range(3):  # the first filtered calls size their workspaces
                ix.search_batch(queries[0], a.k, a.search_size, filters=(off, ids))
                torch.cuda.synchronize()
            ix.profile_read()
            t0 = time.perf_counter()
            for b in range(1, 4):
                ix.search_batch(queries[b % queries.shape[0]], a.k, a.search_size, filters=(off, ids))
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            kms = float(np.mean(ix.profile_read()))
            out["filtered_walk_%d_ids" % size] = {"kernel_ms": round(kms, 4), "kernel_qps": round(nq / kms * 1e3, 1),
                                                  "call_ms": round(dt * 1e3, 2), "call_qps": round(nq / dt, 1)}
    finally:
        ix.set_profiling(False)
    return out


def latency_points(a, ix, queries):
    """A REST request is ONE query (IndexVamana.Search, vamana.go:278-310): whole sdb_index_search_batch calls of 1 .. 256
    queries, device-resident in and out, median of 30.  `workgroup_per_query` is what the library does for calls of up to
    256 queries (16 waves per query, wave 0 walks, all split every hop's rows; the next hop's adjacency row fetched ahead);
    `one_wave_per_query` is the batch kernel forced onto the same calls (SDB_TUNE_WIDE_WALK = 1)."""
    flat_q = queries.reshape(-1, queries.shape[-1])
    out = {}
    try:
        for mode, name in ((0, "workgroup_per_query"), (1, "one_wave_per_query")):
            ix.set_tuning("wide_walk", mode)
            res = {}
            for nq in (1, 16, 64, 256):
                reps = 30
                qs = [flat_q[(i * nq) % (flat_q.shape[0] - nq):(i * nq) % (flat_q.shape[0] - nq) + nq].contiguous()
                      for i in range(reps + 3)]
                for i in range(3):
                    ix.search_batch(qs[i], a.k, a.search_size)
                ts = []
                for i in range(reps):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ix.search_batch(qs[3 + i], a.k, a.search_size)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t0) * 1e3)
                res[str(nq)] = round(float(np.median(ts)), 4)
            out[name] = res
        # the REST path end to end: query in host memory, answer in host memory, one blocking SDB_MEM_HOST call
        ix.set_tuning("wide_walk", 0)
        host_q = flat_q[:4096].cpu().numpy()
        res = {}
        for nq in (1, 16, 64, 256):
            ts = []
            for i in range(33):
                qn = host_q[(i * nq) % (4096 - nq):(i * nq) % (4096 - nq) + nq]
                t0 = time.perf_counter()
                ix.search_batch(qn, a.k, a.search_size)
                if i >= 3:
                    ts.append((time.perf_counter() - t0) * 1e3)
            res[str(nq)] = round(float(np.median(ts)), 4)
        out["host_memory_call"] = res
    finally:
        ix.set_tuning("wide_walk", 0)
    return out


def dependent_fetch_ns(buf, walks):
    """one dependent HBM round trip as a lone lane sees it (tools/probe chase_probe): a pointer chase over `buf`,
    unloaded (one wave) and loaded (one chasing wave per walk of a batch).  None when the probe library is not there."""
    try:
        lib = C.CDLL(os.path.join(ROOT, "tools", "probe", "libgather_probe.so"))
        lib.chase_probe.restype = C.c_double
        lib.chase_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        nbytes = buf.numel() * buf.element_size()
        torch.cuda.synchronize()
        one = lib.chase_probe(buf.data_ptr(), nbytes, 2000, 1, None)
        many = lib.chase_probe(buf.data_ptr(), nbytes, 2000, int(walks), None)
        if one <= 0 or many <= 0:
            return None
        return {"unloaded": round(one, 1), "loaded": round(many, 1), "waves_loaded": int(walks),
                "over_bytes": int(nbytes)}
    except Exception as e:  # measurement aid only
        log("dependent-fetch probe unavailable: %r" % (e,))
        return None


def c4_point(a, dev, dev_index):
    """BASELINE configs[3] (vectorVamana + product quantizer, d = 768, K = 256) inside the default line, at --c4-rows
    rows (1M by default so that the driver's run stays short; `--config c4` is the full 10M x 768 run).  Per M: whole-call
    QPS (LUT build + walk), kernel QPS, recall@10 without re-ranking (like the reference), SURVEY 8d's K5 bytes
    (n_dist * M code bytes + edge ids) over the kernel time."""
    from semadb_amd import vectorstore as vs
    d, n, nq, k, L = 768, a.c4_rows, a.batch, a.k, a.search_size
    base = gen_rows(n, d, 20250620, a.dist, dev)
    nbq = 6
    queries = gen_rows(nbq * nq, d, 20250621, a.dist, dev).view(nbq, nq, d)

    class P:
        metric, search_size, degree_bound, alpha = a.metric, a.search_size, a.degree_bound, a.alpha
    ix, build_s = build_index(P, base, dev_index, name="c4")
    truth = torch.cat([exact_topk(queries[b], base, k)[1] + 2 for b in range(nbq)])
    ix.set_profiling(True)

    def measure():
        for b in range(2):
            ix.search_batch(queries[b], k, L)
        torch.cuda.synchronize()
        ix.profile_read()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(3):
            for b in range(nbq):
                ix.search_batch(queries[b], k, L)
        e1.record()
        torch.cuda.synchronize()
        call_ms = e0.elapsed_time(e1) / (3 * nbq)
        kms = float(np.mean(ix.profile_read()))
        hits = nd = ne = nh = 0
        for b in range(nbq):
            ids, _, _, tr = ix.search_batch(queries[b], k, L, trace=True)
            hits += int((ids.to(torch.int64).unsqueeze(2) == truth[b * nq:(b + 1) * nq].unsqueeze(1)).any(2).sum().item())
            nd += int(tr.n_dist.to(torch.int64).sum().item())
            ne += int(tr.n_edges.to(torch.int64).sum().item())
            nh += int(tr.n_hop.to(torch.int64).sum().item())
        return {"call_qps": round(nq / call_ms * 1e3, 1), "call_ms": round(call_ms, 4), "kernel_ms": round(kms, 4),
                "kernel_qps": round(nq / kms * 1e3, 1), "recall_at_10": round(hits / (nbq * nq * k), 4),
                "n_dist_per_batch": nd / nbq, "n_edges_per_batch": ne / nbq, "n_hop_per_batch": nh / nbq}

    dep_ns = dependent_fetch_ns(base, nq)
    out = {"workload": "vectorVamana + product quantizer %dx%d %s, K=256, searchSize=%d degreeBound=%d, batch=%d" %
                       (n, d, a.metric, L, a.degree_bound, nq), "build_s": round(build_s, 2),
           "recall_note": "no re-ranking, like the reference (product.go:238-277): recall is the quantizer's"}
    full = measure()
    full["GB/s"] = round((full["n_dist_per_batch"] * d * 4 + full["n_edges_per_batch"] * 4) / full["kernel_ms"] / 1e6, 1)
    out["full_precision"] = full
    # the full-precision walk with the two-precision hop (SDB_TUNE_SKETCH; a float16 copy of the rows beside them): the same
    # answers -- compared on every batch -- without a quantizer's loss of recall
    try:
        ref = [tuple(t.clone() for t in ix.search_batch(queries[b], k, L)[:3]) for b in range(nbq)]
        ix.set_tuning("sketch", 1)
        if ix.sketch_stats()[2]:
            same = True
            for b in range(nbq):
                got = ix.search_batch(queries[b], k, L)[:3]
                same &= bool(torch.equal(got[0], ref[b][0]) and torch.equal(got[1].view(torch.int32), ref[b][1].view(torch.int32)) and
                             torch.equal(got[2], ref[b][2]))
            tp = measure()
            out["full_precision_two_precision_hop"] = {"call_qps": tp["call_qps"], "call_ms": tp["call_ms"], "kernel_ms": tp["kernel_ms"],
                                                       "recall_at_10": tp["recall_at_10"], "identical_to_the_default_walk": same}
        else:
            out["full_precision_two_precision_hop"] = {"note": "no room for the float16 copy"}
    except Exception as e:
        out["full_precision_two_precision_hop"] = {"error": repr(e)}
    finally:
        ix.set_tuning("sketch", 0)
    keep = []
    for M in [int(x) for x in a.pq_m.split(",")]:
        if d % M:
            continue
        train_n = min(10000, n)
        train = base[:train_n].cpu().numpy().copy()
        pq = vs.ProductQuantizer(a.metric, vs.ProductQuantizerParameters(256, M, train_n), d, device=dev_index)
        torch.cuda.synchronize()
        t0 = time.time()
        pq.Fit(train, np.arange(M) * 7 % train_n, alias=True)  # product.go:175-236: M k-means fits, all at once
        fit_s = time.time() - t0
        t0 = time.time()
        vs.attach(ix, pq)  # encodes every stored row (product.go:136-159)
        torch.cuda.synchronize()
        enc_s = time.time() - t0
        m = measure()
        enc_tflops = (n + 1) * d * 256 * 2.0 / enc_s / 1e12  # assignment is FP32-vector bound, 2 K flops per float
        m["fit_s"], m["encode_s"] = round(fit_s, 3), round(enc_s, 3)
        # the encode KERNEL on its own (sdb_pq_encode over 2M resident rows, HIP events): the attach call above also
        # allocates the code table and writes the neighbours' code rows, which hid the kernel's own fraction
        try:
            sub = base[:min(n, 2_000_000)]
            pq.encode(sub[:1024])
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            pq.encode(sub)
            ev1.record()
            torch.cuda.synchronize()
            k_ms = ev0.elapsed_time(ev1)
            k_tf = sub.shape[0] * d * 256 * 2.0 / (k_ms * 1e-3) / 1e12
            m["encode_kernel_roofline"] = {"bound": "fp32_vector", "kernel": "k_pq_encode_t" if M <= 24 else "k_pq_encode_pair",
                                           "rows": int(sub.shape[0]), "ms": round(k_ms, 3), "achieved": round(k_tf, 2),
                                           "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                           "frac": round(k_tf / FP32_VECTOR_PEAK_TFLOPS, 4)}
        except Exception as e:
            m["encode_kernel_roofline"] = {"error": repr(e)}
        m["encode_roofline"] = {"bound": "fp32_vector", "kernel": "k_pq_encode_t", "achieved": round(enc_tflops, 2),
                                "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(enc_tflops / FP32_VECTOR_PEAK_TFLOPS, 4),
                                "note": "whole attach call: encode of all rows + code upload bookkeeping"}
        alg = m["n_dist_per_batch"] * M + m["n_edges_per_batch"] * 4
        m["traffic_over_algorithmic"] = None  # PMC pass (tools/pmc_c4.sh) only; not measured inside this run
        try:  # the committed PMC pass of the same kernels at 1M x 768: profile-derived, labelled as such
            rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_c4.json")))
            if "M=%d" % M in rec:
                m["traffic_over_algorithmic"] = rec["M=%d" % M]["traffic_over_codes_and_edges"]
                m["traffic_source"] = "profile-derived, not measured in this run: " + rec["source"]
        except Exception:
            pass
        hbm_frac = round(alg / m["kernel_ms"] / 1e6 / HBM_PEAK_GBS, 4)
        m["roofline"] = {"bound": "hbm", "kernel": pq_walk_kernel(M),
                         "achieved": round(alg / m["kernel_ms"] / 1e6, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                         "traffic": None}
        # The roof that binds a quantized walk is not bytes: a hop cannot start before the previous hop has named its node,
        # so a query's walk takes at least hops x (dependent fetches per hop) x one dependent HBM round trip -- measured here,
        # on this box, by a one-lane pointer chase over the same 10M x 768 slab (tools/probe chase_probe; unloaded and
        # with one chasing lane per walk of a batch).  latency_roof.frac = that floor / the kernel's time: the share of
        # the batch's duration that is irreducible memory latency; the rest is the wave's own instruction stream.
        fetches = 1 if M <= 32 else 2  # M <= 32: adjacency row + code rows in one fetch; else the code gather follows the row
        hops = m["n_hop_per_batch"] / nq
        if dep_ns:
            floor_ms = hops * fetches * dep_ns["loaded"] * 1e-6
            m["latency_roof"] = {"bound": "dependent-fetch latency", "hops_per_query": round(hops, 1),
                                 "dependent_fetches_per_hop": fetches, "dependent_fetch_ns": dep_ns,
                                 "floor_ms": round(floor_ms, 4), "kernel_ms": m["kernel_ms"],
                                 "frac": round(floor_ms / m["kernel_ms"], 4),
                                 "hop_us": round(m["kernel_ms"] * 1e3 / hops, 2)}
            if m["latency_roof"]["frac"] > hbm_frac:  # what binds goes into the roofline object's own words
                m["roofline"]["bound"] = "latency"
                m["roofline"]["bound_note"] = ("hops x dependent-fetch latency: latency_roof.frac %.3f of the kernel's time "
                                               "against %.3f of the HBM roof" % (m["latency_roof"]["frac"], hbm_frac))
        if M <= 32:  # what the one-fetch-per-hop layout costs in HBM (counted in sdb_index_size_in_memory)
            m["code_row_layout_bytes"] = int(2 * 64 * M * (n + 1))
        if M <= 32:
            # the neighbours' code rows sit behind the adjacency row (index.h d_adjcodes): a hop FETCHES 256 B of ids and
            # 64 M B of codes whatever the visited set then says -- the model of the bytes requested, beside the algorithmic ones
            fetched = m["n_hop_per_batch"] * (256 + 64 * M)
            m["fetched_bytes_model_over_algorithmic"] = round(fetched / alg, 3)
            m["layout"] = "neighbour code rows behind the adjacency row: one fetch per hop (node.go:37-54)"
            if M * 256 <= 2048:
                m["bound_by"] = ("one query's instruction stream, not bytes: two waves per query (the walker fetches, tests the visited "
                                 "set, sums and names the next node; the merger runs AddWithLimit), a 1 024-query batch is 4 + 4 waves per CU")
        else:
            # the table look-ups, not bytes, are this kernel's work: M per distance, from LDS (ds_read_b32) or from the
            # register tables (four ds_bpermute + selects each).  The guide's aggregate ds_read_b32 rate is ~75 TB/s =
            # 18.75 T look-ups/s: the walk is far below it and far below HBM -- what binds is latency at low occupancy
            lookups = m["n_dist_per_batch"] * M / (m["kernel_ms"] * 1e-3)
            m["lookups_per_s"] = round(lookups / 1e12, 3)
            m["lookups_unit"] = "T table look-ups/s (M x n_dist / kernel time)"
            m["frac_of_lds_read_rate"] = round(lookups / LDS_B32_READS_PER_S, 4)
            m["bound_by"] = ("latency at two walks per CU: a distance is M dependent look-ups (ds_read_b32 ~50 cycles issue to "
                             "use; 132 of 192 tables sit in registers and cost four ds_bpermute each) summed in index order "
                             "across the four waves' turns; neither HBM (roofline.frac) nor the LDS array "
                             "(frac_of_lds_read_rate) is near its limit")
        out["M=%d" % M] = m
        log("c4 point M=%d: %s" % (M, json.dumps(m)))
        keep.append(pq)
    ix.close()
    for pq in keep:
        pq.close()
    del base
    torch.cuda.empty_cache()
    return out


def host_rates(a, ix, queries, k, L, last_device_result):
    """SURVEY 8d protocol: QPS including H2D of the queries and D2H of the results.
    host_qps     one sdb_index_search_batch(SDB_MEM_HOST) call per batch from pinned host memory
    batcher_qps  single-query Search calls from many threads through the C++ mirror's micro-batcher"""
    from semadb_amd import _lib
    nb, nq, d = queries.shape
    out = {}
    qh = torch.empty((nb, nq, d), dtype=torch.float32, pin_memory=True)
    qh.copy_(queries)
    q_np = qh.numpy()
    o_ids = torch.empty((nq, k), dtype=torch.int64, pin_memory=True).numpy().view(np.uint64)
    o_d = torch.empty((nq, k), dtype=torch.float32, pin_memory=True).numpy()
    o_c = torch.empty((nq,), dtype=torch.int32, pin_memory=True).numpy().view(np.uint32)
    for b in range(3):
        ix.search_batch(q_np[b], k, L, out=(o_ids, o_d, o_c))
    t0 = time.perf_counter()
    for b in range(nb):
        ix.search_batch(q_np[b], k, L, out=(o_ids, o_d, o_c))
    dt = time.perf_counter() - t0
    out["host_qps_via_python"] = round(nb * nq / dt, 1)  # the same calls through ctypes: + the interpreter between them
    # the batcher: libsemadb_hostbench.so (semadb_amd/host/hostbench.cpp over semadb_host.hpp's SearchBatcher)
    so = os.path.join(ROOT, "semadb_amd", "libsemadb_hostbench.so")
    hb = C.CDLL(so)
    # host_qps: the blocking calls issued from C (what a Go host pays through cgo), page-locked slabs of sdb_host_alloc:
    # the walk reads the queries and writes the results in place, one launch and one wait per call
    hb.sdb_hostbench_blocking.restype = C.c_int
    hb.sdb_hostbench_blocking.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                          C.POINTER(C.c_double)]
    flat_q = np.ascontiguousarray(q_np.reshape(nb * nq, d))
    blk = {}
    for label, pinned, no_zc in (("in_place", 1, 0), ("staged", 1, 1), ("pageable", 0, 0)):
        ix.set_tuning("no_zero_copy", no_zc)
        qps_b, ms_b = C.c_double(0), C.c_double(0)
        b_ids = np.zeros((nq, k), dtype=np.uint64)
        b_c = np.zeros(nq, dtype=np.uint32)
        rc_b = hb.sdb_hostbench_blocking(ix._h, d, flat_q.ctypes.data, nb, nq, k, L, 3, pinned, b_ids.ctypes.data, b_c.ctypes.data,
                                         C.byref(qps_b), C.byref(ms_b))
        ids0, _, c0, _ = ix.search_batch(q_np[0], k, L)
        blk[label] = {"qps": round(qps_b.value, 1), "ms_per_batch": round(ms_b.value, 4), "rc": rc_b,
                      "matches_direct_call": bool(np.array_equal(b_ids, ids0) and np.array_equal(b_c, c0))}
    ix.set_tuning("no_zero_copy", 0)
    out["host_qps"] = blk["in_place"]["qps"]
    out["host_ms_per_batch"] = blk["in_place"]["ms_per_batch"]
    out["host_blocking_variants"] = blk
    out["host_qps_definition"] = "queries in page-locked host memory -> sdb_index_search_batch(SDB_MEM_HOST) -> results in " \
                                 "page-locked host memory, one blocking call per batch of %d issued from C (3 x %d batches); the " \
                                 "kernel reads and writes the host buffers in place (host_blocking_variants.staged: the " \
                                 "same with staging copies forced; .pageable: malloc'ed buffers)" % (nq, nb)
    hb.sdb_hostbench_batcher.restype = C.c_int
    hb.sdb_hostbench_batcher.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    first_ids = np.zeros((nb * nq, k), dtype=np.uint64)
    first_c = np.zeros(nb * nq, dtype=np.uint32)
    threads, depth, workers = 64, 48, 4
    qps, batches, served = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    p50, p99 = C.c_double(0), C.c_double(0)
    rc = hb.sdb_hostbench_batcher(ix._h, d, flat_q.ctypes.data, nb * nq, k, L, threads, depth, nq, 300, workers, 2.0,
                                  first_ids.ctypes.data, first_c.ctypes.data, C.byref(qps), C.byref(batches),
                                  C.byref(served), C.byref(p50), C.byref(p99))
    out["batcher_qps"] = round(qps.value, 1)
    out["batcher_vs_host"] = round(qps.value / out["host_qps"], 3)
    out["batcher_latency_us"] = {"p50": round(p50.value, 1), "p99": round(p99.value, 1),
                                 "note": "per request, submit -> answered, at %d requests outstanding" % (threads * depth)}
    out["batcher_definition"] = "%d submitting threads x %d single-query Search requests outstanding each (a Go server's " \
                                "request goroutines), coalesced by semadb_host.hpp SearchBatcher into host-memory " \
                                "batches of <= %d, %d batches in flight; mean device batch %.0f queries; rc %d" % (
                                    threads, depth, nq, workers, served.value / max(1, batches.value), rc)
    # the other end of the load range: 1 and 8 requests outstanding (a lone REST client; a handful).  The batcher seals a
    # partial batch at once when no device batch is running, so a lone request does not wait out the window
    light = {}
    for t_, d_ in ((1, 1), (8, 1)):
        q2, b2, s2, p50b, p99b = C.c_double(0), C.c_uint64(0), C.c_uint64(0), C.c_double(0), C.c_double(0)
        rc2 = hb.sdb_hostbench_batcher(ix._h, d, flat_q.ctypes.data, nb * nq, k, L, t_, d_, nq, 300, workers, 0.5, None, None,
                                       C.byref(q2), C.byref(b2), C.byref(s2), C.byref(p50b), C.byref(p99b))
        light["%d_outstanding" % (t_ * d_)] = {"p50_us": round(p50b.value, 1), "p99_us": round(p99b.value, 1),
                                               "qps": round(q2.value, 1), "mean_batch": round(s2.value / max(1, b2.value), 2),
                                               "rc": rc2}
    out["batcher_light_load"] = light
    # batches in flight: the same load at 2 workers (rounds 3-5's setting) beside the default's 4
    q3, b3, s3, p50c, p99c = C.c_double(0), C.c_uint64(0), C.c_uint64(0), C.c_double(0), C.c_double(0)
    rc3 = hb.sdb_hostbench_batcher(ix._h, d, flat_q.ctypes.data, nb * nq, k, L, threads, depth, nq, 300, 2, 1.0, None, None,
                                   C.byref(q3), C.byref(b3), C.byref(s3), C.byref(p50c), C.byref(p99c))
    out["batcher_qps_2_in_flight"] = round(q3.value, 1) if rc3 == 0 else None
    # the batcher's answers are the same answers: compare the first batch with a direct call
    ids0, _, c0, _ = ix.search_batch(q_np[0], k, L)
    out["batcher_matches_direct_call"] = bool(np.array_equal(first_ids[:nq], ids0) and np.array_equal(first_c[:nq], c0))
    return out


def secondary_points(a, dev, dev_index):
    """The headline sits on embedding-like data (latent:24).  Two more points so the number cannot be read as
    "any 384-d data": a harder set that still passes the gate, and SURVEY 8d's planned i.i.d. Gaussian rows
    (which no graph index can serve at recall 0.95 with an API-legal searchSize)."""
    out = {}
    nq, k, L, d = a.batch, a.k, a.search_size, a.dim
    for dist_name in ("latent:28", "gaussian"):
        base = gen_rows(a.rows, d, 20250620, dist_name, dev)
        queries = gen_rows(6 * nq, d, 20250621, dist_name, dev).view(6, nq, d)
        # (not strict: the sweep below searches beyond the API's searchSize range of 25 .. 75; the build's own parameters
        # are inside it either way)
        ix, build_s = build_index(a, base, dev_index, name="sec", strict=False)
        hits = 0
        for b in range(2):
            ids, _, _, _ = ix.search_batch(queries[b], k, L)
            truth = exact_topk(queries[b], base, k)[1] + 2
            hits += int((ids.to(torch.int64).unsqueeze(2) == truth.unsqueeze(1)).any(2).sum().item())
        for b in range(2, 4):
            ix.search_batch(queries[b], k, L)
        torch.cuda.synchronize()
        ix.set_profiling(True)
        ix.profile_read()
        t0 = time.perf_counter()
        reps = 12
        for r in range(reps):
            ix.search_batch(queries[2 + r % 4], k, L)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kms = [float(v) for v in ix.profile_read()][-reps:]
        ix.set_profiling(False)
        # counter pass (B(q) = n_dist d 4 + edge ids read), the timed batches again with the trace on
        alg = {}
        nd_sum = nh_sum = tail_sum = 0.0
        for b in range(2, 6):
            _, _, _, tr = ix.search_batch(queries[b], k, L, trace=True)
            nd = int(tr.n_dist.to(torch.int64).sum().item())
            ne = int(tr.n_edges.to(torch.int64).sum().item())
            alg[b] = nd * d * 4 + ne * 4
            nd_sum += nd / nq
            nh_sum += float(tr.n_hop.float().mean().item())
            ndq = tr.n_dist.float()  # a batch ends on its longest walk: how far that is from the mean
            tail_sum += float(ndq.max().item() / ndq.mean().item())
        alg_total = sum(alg[2 + r % 4] for r in range(reps))
        ach = alg_total / (sum(kms) * 1e-3) / 1e9 if kms else 0.0
        # the two-precision hop on this dataset too (config.two_precision_hop is the headline's): same answers, its rate
        two_p = {}
        try:
            ref = [ix.search_batch(queries[b], k, L)[:3] for b in range(2, 6)]
            ref = [(r[0].clone(), r[1].clone().view(torch.int32), r[2].clone()) for r in ref]
            ix.set_tuning("sketch", 1)
            same = True
            for i, b in enumerate(range(2, 6)):
                ids, dd, cnt, _ = ix.search_batch(queries[b], k, L)
                same &= bool(torch.equal(ids, ref[i][0]) and torch.equal(dd.view(torch.int32), ref[i][1]) and torch.equal(cnt, ref[i][2]))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(reps):
                ix.search_batch(queries[2 + r % 4], k, L)
            torch.cuda.synchronize()
            two_p = {"two_precision_hop_qps": round(reps * nq / (time.perf_counter() - t0), 1),
                     "two_precision_hop_identical_to_the_default_walk": same}
        except Exception as e:
            two_p = {"two_precision_hop_error": repr(e)}
        finally:
            ix.set_tuning("sketch", 0)
        # the same batches with two of them in flight (what the tail of a batch -- longest_walk_over_mean -- costs)
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for r in range(2):
            with torch.cuda.stream(streams[r]):
                ix.search_batch(queries[2 + r], k, L)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(4 * reps):  # (longer than the one-at-a-time loop: filling and draining the pair is not the point)
            with torch.cuda.stream(streams[r % 2]):
                ix.search_batch(queries[2 + r % 4], k, L)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / 4
        out[dist_name] = {"qps": round(reps * nq / dt, 1), "recall_at_10": round(hits / (2 * nq * k), 4),
                          "build_s": round(build_s, 2), "passes_gate": hits / (2 * nq * k) >= 0.95,
                          "mean_n_dist": round(nd_sum / 4, 1), "mean_n_hop": round(nh_sum / 4, 1),
                          "longest_walk_over_mean": round(tail_sum / 4, 3),
                          "two_batches_in_flight_qps": round(reps * nq / dt2, 1),
                          "two_batches_in_flight_frac": round(alg_total / dt2 / 1e9 / HBM_PEAK_GBS, 4),
                          "algorithmic_bytes_per_launch": int(alg_total / reps),
                          "kernel_ms_avg": round(float(np.mean(kms)), 4) if kms else None,
                          "roofline": {"bound": "hbm", "kernel": "k_greedy_search", "achieved": round(ach, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}}
        out[dist_name].update(two_p)
        if hits / (2 * nq * k) < 0.95:
            # "report the searchSize needed for recall 0.95 separately" -- the device walk takes searchSize up
            # to 512 (the API's maximum is 75, models/search.go:287-297)
            truth0 = exact_topk(queries[0], base, k)[1] + 2
            need, sweep = None, {}
            for L2 in (128, 192, 256, 384, 512):
                ids, _, _, _ = ix.search_batch(queries[0], k, L2)
                r = float((ids.to(torch.int64).unsqueeze(2) == truth0.unsqueeze(1)).any(2).float().mean().item())
                sweep[str(L2)] = round(r, 4)
                if r >= 0.95:
                    need = L2
                    break
            out[dist_name]["recall_by_search_size"] = sweep
            out[dist_name]["search_size_for_recall_0.95"] = need if need is not None else "none <= 512"
        log("secondary %s: %.0f QPS, recall %.4f, build %.2fs" %
            (dist_name, out[dist_name]["qps"], out[dist_name]["recall_at_10"], build_s))
        ix.close()
        del base, queries
        torch.cuda.empty_cache()
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_affinity():
    """what bounds the threads: CPUs visible, the process's affinity mask, the cgroup quota"""
    out = {"cpus_online": os.cpu_count()}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        out["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception:
        pass
    return out


def cpu_baseline(a, ix, queries, k, L):
    """The reference algorithm on the host cores: C restatement (oracle/) with the AVX2 intrinsics
    transcription of distance/asm/dot.s, same graph (exported from HBM), same query batches."""
    from oracle import oracle as orc
    t0 = time.time()
    ids, vecs, offsets, edges = ix.export()
    impl = orc.IMPL_AVX2 if orc.has_avx2() else orc.IMPL_ASM
    o = orc.Index(a.dim, a.metric, a.degree_bound, a.search_size, a.alpha, impl=impl)
    assert o.load(ids, vecs, offsets, edges) == 0
    del vecs, edges
    log("cpu baseline: graph exported and loaded into the oracle in %.1fs" % (time.time() - t0))
    qb = queries.cpu().numpy()
    nb, nq, d = qb.shape
    threads = orc.effective_cpus()  # affinity capped by the cgroup quota
    # all cores, one query per thread (goroutine-per-request); bounded sample = the distinct batches, repeated
    sample = np.concatenate([qb.reshape(nb * nq, d)] * a.cpu_repeat)
    o.search_batch(sample[:nq], k, L, n_threads=threads)  # warm
    t1 = time.perf_counter()
    c_ids, c_d, c_cnt, c_nd, c_nh, c_ne = o.search_batch(sample, k, L, n_threads=threads)
    t_all = time.perf_counter() - t1
    # one thread, for comparison with the reference README's single-thread table
    n1 = min(512, nb * nq)
    t2 = time.perf_counter()
    o.search_batch(sample[:n1], k, L, n_threads=1)
    t_one = time.perf_counter() - t2
    # parity at full size: GPU results for the same batches
    mism_ids = mism_d = mism_nd = 0
    for b in range(nb):
        g_ids, g_d, g_c, tr = ix.search_batch(queries[b], k, L, trace=True)
        torch.cuda.synchronize()
        gi = g_ids.cpu().numpy().view(np.uint64)
        gd = g_d.cpu().numpy()
        sl = slice(b * nq, (b + 1) * nq)
        mism_ids += int((gi != c_ids[sl]).any(axis=1).sum())
        mism_d += int((gd.view(np.uint32) != c_d[sl].view(np.uint32)).any(axis=1).sum())
        mism_nd += int((tr.n_dist.cpu().numpy().astype(np.uint64) != c_nd[sl]).sum())
    return {
        "value": round(len(sample) / t_all, 1),
        "unit": "queries/s",
        "cores": threads,
        # which host this was: the figure moved 20.2 k -> 14.8 k queries/s between two rounds' boxes with no code change
        "cpu_model": _cpu_model(),
        "affinity": _cpu_affinity(),
        "kind": "port",
        "sample": "%d queries (%d distinct batches of %d, x%d) on the same 1M graph; C restatement of "
                  "greedySearch with AVX2 transcription of distance/asm/dot.s; one query per thread" %
                  (len(sample), nb, nq, a.cpu_repeat),
        "single_thread_qps": round(n1 / t_one, 1),
        "cpu_seconds": round(t_all * threads, 1),
        # parity at full size, as scalars (nested objects do not survive the driver's record): the GPU's answers to the
        # same batches against the oracle's -- result ids, distance bits, distFn evaluation counts
        "parity_queries": nb * nq,
        "parity_id_mismatch_queries": mism_ids,
        "parity_dist_bits_mismatch_queries": mism_d,
        "parity_n_dist_mismatch_queries": mism_nd,
        "parity_full_size": {"queries": nb * nq, "id_mismatch_queries": mism_ids,
                             "dist_bits_mismatch_queries": mism_d, "n_dist_mismatch_queries": mism_nd},
    }


# ------------------------------------------------------------------------------------------------------------
# C3: index build
# ------------------------------------------------------------------------------------------------------------
def run_c3(a, ctx):
    """BASELINE configs[2]: 1M x 384 insert + RobustPrune (alpha 1.2) on one MI355X.  A step = one full build."""
    dev, dev_index = ctx["dev"], ctx["dev_index"]
    d, n = a.dim, a.rows
    base = gen_rows(n, d, 20250620, a.dist, dev)
    steps = max(1, min(a.steps, 5))
    times, roof, degs = [], None, None
    for s in range(min(a.warmup, 1) + steps):
        ix, build_s = build_index(a, base, dev_index, name="c3")
        if s >= min(a.warmup, 1):
            times.append(build_s)
            roof = build_roofline(ix, n, d, build_s)
            nn, ne, _ = ix.stats()
            degs = ne / nn
        if s == min(a.warmup, 1) + steps - 1:  # quality of the graph that was timed
            queries = gen_rows(4 * a.batch, d, 20250621, a.dist, dev).view(4, a.batch, d)
            hits = 0
            for b in range(4):
                ids, _, _, _ = ix.search_batch(queries[b], a.k, a.search_size)
                truth = exact_topk(queries[b], base, a.k)[1] + 2
                hits += int((ids.to(torch.int64).unsqueeze(2) == truth.unsqueeze(1)).any(2).sum().item())
            recall = hits / (4 * a.batch * a.k)
        ix.close()
        torch.cuda.empty_cache()
    t = float(np.mean(times))
    return {
        "metric": "index build 1Mx384 Vamana insert + RobustPrune (alpha=1.2)", "value": round(n / t, 1),
        "unit": "inserts/s", "n_gpus": 1, "steps": steps, "warmup": min(a.warmup, 1),
        "ms_per_step": round(t * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "vectorVamana index build %dx%d %s: insert + RobustPrune alpha=%.1f, searchSize=%d "
                               "degreeBound=%d, device-resident vectors, one MI355X" %
                               (n, d, a.metric, a.alpha, a.search_size, a.degree_bound),
                   "dataset": "%s seed 20250620" % a.dist, "avg_degree": round(degs, 2),
                   "recall_at_10_of_built_graph": round(recall, 4), "build_s_each": [round(x, 3) for x in times]},
        "roofline": roof,
    }


# ------------------------------------------------------------------------------------------------------------
# C4: product-quantized search
# ------------------------------------------------------------------------------------------------------------
def pq_walk_kernel(M, K=256):
    """the kernel index.hip launch_greedy_search takes for an unfiltered quantized search of the reference's searchSize"""
    if M in (128, 192, 256, 384):
        return "k_greedy_search_pqw"  # table in LDS + registers of four / eight waves
    if M * K <= 2048:
        return "k_greedy_search_pq2"  # table in LDS, walker + merger (two waves per query)
    return "k_greedy_search<PQDist>"  # table in LDS (up to 64 KB) or in global memory, one wave per query


def run_c4(a, ctx):
    """BASELINE configs[3]: vectorVamana + product quantizer, 10M x 768 (K = 256, M from --pq-m; M = 8 is the
    documented configuration and the `value`), LUT distance kernel, one MI355X.  The reference does no
    re-ranking, so recall is reported beside every rate -- it is what the quantizer gives, not a gate."""
    from semadb_amd import vectorstore as vs
    dev, dev_index = ctx["dev"], ctx["dev_index"]
    d, n, nq, k, L = a.dim, a.rows, a.batch, a.k, a.search_size
    base = gen_rows(n, d, 20250620, a.dist, dev)
    nbq = 16
    queries = gen_rows(nbq * nq, d, 20250621, a.dist, dev).view(nbq, nq, d)
    ix, build_s = build_index(a, base, dev_index, name="c4")
    log("c4: built %d x %d in %.1fs" % (n, d, build_s))
    truth = torch.cat([exact_topk(queries[b], base, k)[1] + 2 for b in range(nbq)])
    ix.set_profiling(True)

    def measure(batch_mult=1):
        qs = queries.view(-1, d)
        bsz = nq * batch_mult
        nbat = qs.shape[0] // bsz
        for b in range(min(2, nbat)):
            ix.search_batch(qs[b * bsz:(b + 1) * bsz], k, L)
        torch.cuda.synchronize()
        ix.profile_read()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, a.steps // 5)
        e0.record()
        for r in range(reps):
            for b in range(nbat):
                ix.search_batch(qs[b * bsz:(b + 1) * bsz], k, L)
        e1.record()
        torch.cuda.synchronize()
        call_ms = e0.elapsed_time(e1) / (reps * nbat)
        kms = float(np.mean(ix.profile_read()))
        hits = nd = ne = 0
        for b in range(nbat):
            ids, _, _, tr = ix.search_batch(qs[b * bsz:(b + 1) * bsz], k, L, trace=True)
            eq = (ids.to(torch.int64).unsqueeze(2) == truth[b * bsz:(b + 1) * bsz].unsqueeze(1)).any(2)
            hits += int(eq.sum().item())
            nd += int(tr.n_dist.to(torch.int64).sum().item())
            ne += int(tr.n_edges.to(torch.int64).sum().item())
        return {"kernel_ms": round(kms, 4), "kernel_qps": round(bsz / kms * 1e3, 1), "call_ms": round(call_ms, 4),
                "call_qps": round(bsz / call_ms * 1e3, 1), "recall_at_10": round(hits / (nbat * bsz * k), 4),
                "n_dist_per_batch": nd / nbat, "n_edges_per_batch": ne / nbat}

    full = measure()
    full["GB/s"] = round((full["n_dist_per_batch"] * d * 4 + full["n_edges_per_batch"] * 4) / full["kernel_ms"] / 1e6, 1)
    points = {}
    quantizers = []  # kept alive while attached; the index only borrows them
    train_n = 10000  # the reference's largest TriggerThreshold (models/quantizer.go:62)
    for M in [int(x) for x in a.pq_m.split(",")]:
        if d % M:
            continue
        train = base[:train_n].cpu().numpy().copy()
        pq = vs.ProductQuantizer(a.metric, vs.ProductQuantizerParameters(256, M, train_n), d, device=dev_index)
        t0 = time.time()
        pq.Fit(train, np.arange(M) * 7 % train_n, alias=True)
        fit_s = time.time() - t0
        torch.cuda.synchronize()
        t0 = time.time()
        vs.attach(ix, pq)
        torch.cuda.synchronize()
        enc_s = time.time() - t0
        rec = {"fit_s": round(fit_s, 3), "encode_s": round(enc_s, 3),
               "encode_TFLOP/s": round((n + 1) * d * 256 * 2.0 / enc_s / 1e12, 2),
               "encode_frac_of_fp32_vector_peak": round((n + 1) * d * 256 * 2.0 / enc_s / 1e12 / FP32_VECTOR_PEAK_TFLOPS, 4)}
        for mult in (1, 4, 16) if M == 8 else (1,):
            m = measure(mult)
            # K5 bytes: n_dist * M code bytes + edge ids; the LUT lookups are LDS traffic, not HBM
            m["code_GB/s"] = round((m["n_dist_per_batch"] * M + m["n_edges_per_batch"] * 4) / m["kernel_ms"] / 1e6, 1)
            m["lds_lookup_Glookups/s"] = round(m["n_dist_per_batch"] * M / m["kernel_ms"] / 1e6, 1)
            rec["batch_%d" % (nq * mult)] = m
        points["M=%d" % M] = rec
        log("c4 M=%d: %s" % (M, json.dumps(rec)))
        quantizers.append(pq)
    ix.close()
    for pq in quantizers:
        pq.close()
    # value: the operating point that is worth running -- M = 192 (recall@10 0.83 without re-ranking, faster than the
    # full-precision walk); M = 8, the reference documentation's example, stays in the line (recall 0.1)
    head_m = "M=192" if "M=192" in points else ("M=8" if "M=8" in points else next(iter(points)))
    head = points[head_m]["batch_%d" % nq]
    m_head = int(head_m.split("=")[1])
    kernel = pq_walk_kernel(m_head)
    res = {
        "metric": "QPS, vectorVamana + product quantizer %dMx%d (K=256, %s), PQ-LUT distance kernel, batch=1024" %
                  (n // 1000000, d, head_m) if n >= 1000000 else
                  "QPS, vectorVamana + product quantizer %dx%d (K=256, %s), PQ-LUT distance kernel, batch=1024" % (n, d, head_m),
        "value": head["call_qps"], "unit": "queries/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": head["call_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # the reference does not re-rank, so there is no recall gate on this configuration; the recall of the operating
        # point `value` is quoted on rides at top level, and M = 8 (the documentation's example) stays comparable
        # from round to round under value_m8
        "recall_at_10": head["recall_at_10"],
        "value_m8": points["M=8"]["batch_%d" % nq]["call_qps"] if "M=8" in points else None,
        "recall_at_10_m8": points["M=8"]["batch_%d" % nq]["recall_at_10"] if "M=8" in points else None,
        "config": {"workload": "vectorVamana + product quantizer %dx%d %s, K=256, searchSize=%d degreeBound=%d, batch=%d; "
                               "value = whole call (LUT build + walk) at %s" % (n, d, a.metric, L, a.degree_bound, nq, head_m),
                   "dataset": "%s seed 20250620" % a.dist, "build_s": round(build_s, 2),
                   "recall_note": "no re-ranking, like the reference (product.go:238-277): recall is the quantizer's",
                   "full_precision": full, "quantized": points},
        "roofline": {"bound": "hbm", "kernel": kernel,
                     "achieved": head["code_GB/s"],
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(head["code_GB/s"] / HBM_PEAK_GBS, 4),
                     "traffic": None,  # PMC pass only (tools/pmc_c4.sh -> profiles/*pmc_c4*); nothing measured in this run is quoted
                     "note": "a dependent chain of ~80 hops per query over M-byte code rows: neither HBM nor LDS is "
                             "saturated; the lever is walks in flight per CU -- tables in LDS and in the register files of "
                             "four waves, two queries per CU (search_kernel.h PQWideDist)"},
    }
    return res


if __name__ == "__main__":
    main()
