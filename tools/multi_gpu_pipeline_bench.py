#!/usr/bin/env python3
"""One PROCESS, N GPUs: the database uploaded and indexed once, replicated device to device (sylph_db_replicate), one router pipeline
over the replicas (sylph_pipeline_create_multi) — what `sylph-hip profile --gpus N` runs — fed with resident 1 Gbp samples dealt
round-robin to the GPUs.  Prints ONE JSON line (Gbp/s sketched + profiled, the whole process).  No torch.distributed, no collective.

    python tools/multi_gpu_pipeline_bench.py --gpus 8                 # an 8-GPU node
    python tools/multi_gpu_pipeline_bench.py --gpus 2 --share         # a one-GPU box: two replicas on device 0 (functional check + overhead)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import sylph_amd as S  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="replicas (0: every GPU the process sees)")
    ap.add_argument("--share", action="store_true", help="replicas beyond the device count wrap around (one-GPU boxes)")
    ap.add_argument("--workload", default="c3", choices=sorted(B.WORKLOADS))
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--workers", type=int, default=3)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--seed", type=int, default=20250711)
    ap.add_argument("--dedup-fpr", type=float, default=0.0, help="> 0: sylph's default pair dedup (the cuckoo filter) instead of the exact set")
    a = ap.parse_args()
    n_dev = torch.cuda.device_count()
    n = a.gpus or n_dev
    if n > n_dev and not a.share:
        raise SystemExit(f"{n} replicas asked for, {n_dev} GPU(s) visible (use --share to let them wrap around)")
    devs = [i % n_dev for i in range(n)]
    c, k, read_len = 200, 31, 150
    n_pairs = B.WORKLOADS[a.workload][0]
    torch.cuda.set_device(0)
    dev0 = torch.device("cuda", 0)
    ctxs = [S.Context(d) for d in devs]
    t0 = time.time()
    db0, n_total, community, dbstats, _ = B.build_database(ctxs[0], dev0, a.workload, c, k, a.seed, 0, 1, "replicate")
    t1 = time.time()
    dbs = [db0] + [db0.replicate(ctxs[i]) for i in range(1, n)]
    for cx in ctxs:
        cx.synchronize()
    t2 = time.time()
    # four distinct read sets, resident on every GPU that has a replica
    sets = []
    for i in range(4):
        bases, rec_off = synth.paired_reads(community, n_pairs, read_len=read_len, seed=a.seed + 1_000_003 + 7919 * i)
        sets.append((bases, rec_off))
    torch.cuda.synchronize()
    per_dev = {}
    for d in sorted(set(devs)):
        per_dev[d] = [(b.to(torch.device("cuda", d)), o.to(torch.device("cuda", d))) for b, o in sets] if d != 0 else sets
    for d in per_dev:
        torch.cuda.synchronize(d)
    n_bases = int(sets[0][1][-1].item())
    p = S.Pipeline(dbs, c=c, k=k, paired=True, n_workers=a.workers, depth=a.depth, max_batch=8)
    if a.dedup_fpr:
        p.set_option("dedup_fpr", a.dedup_fpr)
    cap = a.depth * n
    counts = [0] * n

    def run(n_samples):
        sub = done = 0
        while done < n_samples:
            while sub < n_samples and p.outstanding < cap:
                d = devs[sub % n]
                b, o = per_dev[d][sub % 4]
                if not p.submit_device([(b.data_ptr(), o.data_ptr(), 2 * n_pairs, n_bases)], tag=sub):
                    break
                sub += 1
            r = p.next(views=False)
            assert r["tag"] == done
            counts[r["replica"]] += 1
            done += 1

    run(4 * cap)                                     # pools, code objects, first-use allocations of every replica's contexts
    t_a = time.perf_counter()
    run(cap)
    est = (time.perf_counter() - t_a) / cap
    n_timed = max(cap, int(a.seconds / max(est, 1e-6)))
    counts[:] = [0] * n
    for d in per_dev:
        torch.cuda.synchronize(d)
    t_a = time.perf_counter()
    run(n_timed)
    dt = time.perf_counter() - t_a
    out = {"metric": "read Gbp/s sketched + profiled, ONE process over N replicas of the database", "value": round(n_timed * n_bases / 1e9 / dt, 2), "unit": "Gbp/s",
           "n_replicas": n, "devices": devs, "n_gpus_visible": n_dev, "samples_timed": n_timed, "ms_per_sample": round(dt / n_timed * 1e3, 4),
           "genome_comparisons_per_s": round(n_timed * n_total / dt, 1), "samples_per_replica": counts,
           "config": {"workload": B.DESCR[a.workload], "parallelism": f"one process, {n} replica(s) of the database on device(s) {sorted(set(devs))} "
                      f"(sylph_db_replicate: index copied device to device in {t2 - t1:.2f} s after {t1 - t0:.1f} s of generate + upload + index), one router pipeline "
                      f"(sylph_pipeline_create_multi: {a.workers} sketch workers + 1 profile thread per replica, {a.depth} samples in flight each), samples dealt round-robin, "
                      "results in submission order; no collective",
                      "dedup": f"filter, --fpr {a.dedup_fpr}" if a.dedup_fpr else "exact (--fpr 0 semantics)"},
           "index_gb_per_replica": dbstats["index_gb"], "replicate_s": round(t2 - t1, 3)}
    print(json.dumps(out), flush=True)
    p.close()
    for d in dbs[1:]:
        d.close()
    db0.close()
    for cx in ctxs:
        cx.close()


if __name__ == "__main__":
    main()
