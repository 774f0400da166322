#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X sketch + profile hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its N ranks itself;
                                                            under `python -m torch.distributed.run` it is one of them)

Metric (BASELINE.json): read Gbp/s sketched + genome-comparisons/s profiled, 1 Gbp of 2x150 bp reads vs a GTDB-R220-scale
database (113,104 genome sketches, k=31, c=200).  One *step* = one pass of the hot path over the step's samples on every
GPU: sketch each sample (seeding -> exact dedup/count; reads already resident in HBM) and profile the resulting tables against
the resident database (containment counts + coverage vectors back on the host).  `value` = whole-job read Gbp/s through
both stages; the per-stage rates are reported next to it.  Steps are pipelined the way a multi-sample `sylph profile` run
is: sketching runs on worker threads (own context + stream each), profiling on the main thread's context, and the samples of
the next steps are sketched while the current step is profiled (--pipeline-depth; all K steps complete inside the timed
region).  `one_step_at_a_time` repeats the steps strictly one after the other: step latency, and every kernel alone on the GPU.

Workloads (--workload): c3 = BASELINE configs[2], the configuration the metric is quoted on (default at N = 1: one 1 Gbp
sample per step, database on the one GPU); c4 = configs[3] (default at N > 1: 8 samples per GPU per step, database sharded
by k-mer range over the N GPUs, RCCL exchange inside sylph_db_contain_batch_sharded); c2 / c5 = configs[1] / [4];
c3r = c3 with ragged 35-151 bp reads and 0.1 % N (reported beside c3, not instead of it).

Multi-GPU: samples are independent units (no collective in the sketch stage).  --db-mode shard (default for
N > 1, what north_star describes): every rank holds the postings of one k-mer range; per step the library all-gathers the
slice boundaries, sends every rank its 1/N slice of every table (all-to-all), probes, and sends every hit to the rank that
owns its sample (a second all-to-all); --db-mode replicate: every rank holds the whole index (22-38 GB of 288 GB) and no data-path collective
is needed at all.  scaling = weak (per-GPU work fixed).

After the timed region (untimed): --verify compares the containment results of the last step's first sample, for the
sequence-backed genomes + a sample of the decoys, against the CPU oracle on the same table; the CPU baseline leg (rank 0,
N = 1) times the oracle — the C++ restatement of the reference's AVX2/rayon path — on a bounded sample of the same inputs.
These two legs are the only places bench.py touches oracle/.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sylph_amd as S  # noqa: E402
from sylph_amd import shard as SH  # noqa: E402
from sylph_amd import synth  # noqa: E402

WORKLOADS = {
    # name: (n_pairs, n_community, n_seq_backed, n_genomes_total, genome_len, samples per GPU per step, distinct read sets)
    "c3": (3_333_334, 100, 1000, 113_104, 5_000_000, 1, 4),   # BASELINE configs[2]: 1 Gbp vs GTDB-R220-scale DB
    "c3r": (3_333_334, 100, 1000, 113_104, 5_000_000, 1, 4),  # the same with ragged reads + N
    "c4": (3_333_334, 100, 1000, 113_104, 5_000_000, 8, 8),   # BASELINE configs[3]: 64 x 1 Gbp over 8 GPUs = 8 samples per GPU
    "c2": (3_333_334, 100, 1000, 1000, 5_000_000, 1, 4),      # BASELINE configs[1]: 1 Gbp vs 1,000 x 5 Mbp genomes
    "small": (100_000, 8, 24, 2000, 400_000, 2, 3),           # quick functional run
    "c5": (None, 100, 1000, 113_104, 5_000_000, 1, 1),        # BASELINE configs[4]: ONT-like long reads (N50 10 kb, 5 Gbp), c=100
}
DESCR = {"c3": "1 Gbp synthetic 2x150 bp reads vs GTDB-R220-scale DB (113,104 genome sketches), k=31 c=200 (BASELINE configs[2])",
         "c3r": "configs[2] with ragged reads: 2 x 35-151 bp (uniform), 0.1 % N — the short-read kernel without its best case",
         "c4": "64 x 1 Gbp samples per 8 GPUs (8 samples per GPU per step) vs GTDB-R220-scale DB sharded over the GPUs (BASELINE configs[3])",
         "c2": "1 Gbp synthetic 2x150 bp reads vs 1,000 synthetic 5 Mbp genomes, k=31 c=200 (BASELINE configs[1])",
         "small": "functional smoke workload (NOT a BASELINE config)",
         "c5": "ONT-like long reads (N50 10 kb, 5 Gbp, 5 % substitutions) sketched at c=100 vs GTDB-R220-scale DB at c=200 (BASELINE configs[4])"}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def self_launch(args):
    """--gpus N without a launcher: start the N ranks here (what `torch.distributed.run --nproc-per-node N` would do) and
    pass rank 0's JSON line through."""
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    sys.exit(rc)


def build_database(ctx, device, wl, c, k, seed, rank, world, db_mode):
    """-> (Database (shard or whole), n_genomes_total, community genomes tensor, stats, verify set)"""
    n_pairs, n_comm, n_seq, n_total, glen = WORKLOADS[wl][:5]
    t0 = time.time()
    n_mut = n_seq // 10
    community = synth.random_genomes(n_comm, glen, device, seed, mutated_frac=0.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 17)
    sub = torch.tensor([67, 71, 84, 65], dtype=torch.uint8, device=device)   # A->C, C->G, G->T, T->A
    lut = torch.zeros(256, dtype=torch.uint8, device=device)
    lut[torch.tensor([65, 67, 71, 84], device=device)] = sub
    # Database build: sequence-backed genomes (community + unrelated + 10 % mutated copies at 97 % identity) are
    # generated on the device in batches of <= 1 Gbp and sketched by ONE sylph_sketch_genomes call per batch
    per_batch = max(1, min(n_seq, (1 << 30) // glen))
    sk_parts, sk_lens, t_sketch = [], [], 0.0
    for g0 in range(0, n_seq, per_batch):
        g1 = min(n_seq, g0 + per_batch)
        batch = torch.empty((g1 - g0, glen), dtype=torch.uint8, device=device)
        for g in range(g0, g1):
            if g < n_comm:
                seq = community[g]
            elif g >= n_seq - n_mut:
                src = community[(g - (n_seq - n_mut)) % n_comm]
                mask = torch.rand(glen, generator=gen, device=device) < 0.03
                seq = torch.where(mask, lut[src.long()], src)
            else:
                seq = synth.random_genomes(1, glen, device, seed + 1000 + g, mutated_frac=0.0)[0]
            batch[g - g0].copy_(seq)
        torch.cuda.synchronize()
        coff = np.arange(g1 - g0 + 1, dtype=np.uint64) * np.uint64(glen)     # one contig per genome
        goff_b = np.arange(g1 - g0 + 1, dtype=np.uint64)
        ts = time.perf_counter()
        km, koff, _, _ = ctx.sketch_genomes(None, coff, goff_b, c=c, k=k, device_ptr=batch.data_ptr())
        t_sketch += time.perf_counter() - ts
        sk_parts.append(km)
        sk_lens.append(np.diff(koff.astype(np.int64)))
        del batch
    t1 = time.time()
    seq_k = np.concatenate(sk_parts) if sk_parts else np.zeros(0, dtype=np.uint64)
    seq_off = np.zeros(n_seq + 1, dtype=np.int64)
    seq_off[1:] = np.cumsum(np.concatenate(sk_lens)) if sk_lens else 0
    n_decoy = n_total - n_seq
    if n_decoy > 0:
        dk, doff = synth.decoy_sketches(n_decoy, c=c, device=device, seed=seed + 7)
    else:
        dk, doff = torch.zeros(0, dtype=torch.int64, device=device), torch.zeros(1, dtype=torch.int64, device=device)
    # verify set: every sequence-backed genome + 2,000 decoys spread over the id range (genome-major, on the host)
    vd = np.unique(np.linspace(0, max(n_decoy - 1, 0), num=min(2000, n_decoy)).astype(np.int64)) if n_decoy else np.zeros(0, np.int64)
    doff_h = doff.cpu().numpy()
    v_parts = [seq_k] + [dk[int(doff_h[g]):int(doff_h[g + 1])].cpu().numpy().view(np.uint64) for g in vd]
    v_ids = np.concatenate([np.arange(n_seq), n_seq + vd])
    v_off = np.zeros(len(v_ids) + 1, dtype=np.uint64)
    v_off[1:] = np.cumsum(np.concatenate([np.diff(seq_off), np.diff(doff_h)[vd]]))
    verify_set = (v_ids, np.concatenate(v_parts), v_off)
    kmers = torch.cat([torch.from_numpy(seq_k.view(np.int64)).to(device), dk])
    goff = torch.cat([torch.from_numpy(seq_off).to(device), doff[1:] + int(seq_off[-1])])
    del dk
    torch.cuda.synchronize()
    t2 = time.time()
    if db_mode == "shard":      # every rank generated the same database and keeps the postings of its k-mer range
        bounds = S.shard_bounds((2**64 - 1) // c - 1, world)
        db = S.Database(ctx, kmers.data_ptr(), goff.data_ptr(), device_ptrs=True, n_genomes=n_total, shard=(bounds, world, rank))
    else:
        db = S.Database(ctx, kmers.data_ptr(), goff.data_ptr(), device_ptrs=True, n_genomes=n_total)
    ctx.synchronize()
    t3 = time.time()
    stats = dict(n_genomes=int(n_total), db_kmers_total=int(kmers.numel()), shard_kmers=int(db.n_kmers), index_gb=round(db.index_bytes / 1e9, 2),
                 seq_backed_generate_and_sketch_s=round(t1 - t0, 2), seq_backed_sketch_s=round(t_sketch, 4),
                 db_build_gbp_per_s=round(n_seq * glen / 1e9 / max(t_sketch, 1e-9), 2), db_build_genomes=int(n_seq),
                 generate_s=round(t2 - t1, 2), db_upload_index_s=round(t3 - t2, 2))
    # NB: no torch.cuda.empty_cache() here — returning tens of GB to the driver (hipFree) queues page-table work
    # that stalls this process's GPU queues for 15-40 ms at random moments over the next few hundred ms.
    del kmers, goff
    return db, int(n_total), community, stats, verify_set


def cpu_baseline(bases, rec_off, n_pairs, read_len, c, k, db_sample):
    """Oracle (C++ restatement of the reference CPU path) on a bounded sample, on this box's host cores."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    n_s = min(n_pairs, 2_000_000)
    hb = bases[: n_s * 2 * read_len].cpu().numpy()
    ho = rec_off[: 2 * n_s + 1].cpu().numpy().astype(np.uint64)
    mode = O.MODE_AVX2_FAST if O.lib().orc_has_avx2() else O.MODE_SCALAR
    t = time.perf_counter()
    sk = O.sketch_reads(hb, ho, c=c, k=k, mode=mode, paired=True)
    t_sketch = time.perf_counter() - t
    sketch_gbps = n_s * 2 * read_len / t_sketch / 1e9          # one sample = one thread in the reference (sketch.rs:313,371)
    dbk, dbo = db_sample
    ls = O.LoadedSample(sk["kmers"], sk["counts"])
    _, _, t_probe = ls.probe(dbk, dbo, n_threads=cores)         # genomes in parallel on all cores (contain.rs:284)
    ls.close()
    G = len(dbo) - 1
    return dict(sketch_gbp_per_s=sketch_gbps, comparisons_per_s=G / t_probe, sketch_cores=1, probe_cores=cores,
                sample=f"sketch: first {n_s} read pairs ({n_s * 2 * read_len / 1e6:.0f} Mbp of the 1 Gbp sample) on 1 thread "
                       f"({'AVX2 intrinsics' if mode == O.MODE_AVX2_FAST else 'scalar'}); "
                       f"probe: {G} of the 113,104 genomes ({len(dbk) / 1e6:.0f} M k-mers) on {cores} threads vs the {len(sk['kmers'])}-entry "
                       f"sample table; both rates EXTRAPOLATED linearly to the full workload")


def verify_against_oracle(ctx, res, G, sample_ptrs, verify_set, device):
    """Untimed: the last step's first sample — contain_count and the sorted coverage vector of every genome of the verify set
    (all sequence-backed genomes + 2,000 decoys) against the CPU oracle probing the same table."""
    from oracle import oracle as O
    dk, dc, n = sample_ptrs
    sk = SH.device_view(dk, n, torch.int64, device).cpu().numpy().view(np.uint64)
    sc = SH.device_view(dc, n, torch.int32, device).cpu().numpy().view(np.uint32)
    v_ids, v_k, v_off = verify_set
    ecc, ecov, _ = O.contain(sk, sc, v_k, v_off, n_threads=min(32, os.cpu_count() or 1))
    cc, off, covs = res
    bad = 0
    for j, g in enumerate(v_ids):
        g = int(g)
        got = np.asarray(covs[int(off[g]):int(off[g + 1])]).astype(np.uint32)
        if int(cc[g]) != int(ecc[j]) or not np.array_equal(got, np.sort(ecov[j])):
            bad += 1
    return {"genomes_checked": int(len(v_ids)), "genomes_with_hits": int((ecc > 0).sum()), "hits_checked": int(ecc.sum()), "mismatches": int(bad),
            "what": "contain_count + sorted coverage vector per genome vs the CPU oracle on the same sample table (first sample of the last step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("SYLPH_BENCH_WORKLOAD", "auto"), choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--db-mode", default=os.environ.get("SYLPH_BENCH_DB_MODE", "auto"), choices=["auto", "replicate", "shard"])
    ap.add_argument("--samples-per-step", type=int, default=0, help="samples per GPU per step (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-fed (PCIe-inclusive) leg")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip the in-library HIP-event kernel timers (no roofline objects)")
    ap.add_argument("--sketch-workers", type=int, default=0,
                    help="sketch worker threads, each with its own context/stream (default 2; the reference sketches samples on parallel threads too, sketch.rs:313)")
    ap.add_argument("--pipeline-depth", type=int, default=0,
                    help="steps in flight: the samples of step i+1.. are sketched while step i is profiled (default: workers + 1 for one sample per step, else 2; 1 = one step at a time)")
    ap.add_argument("--no-sequential-leg", action="store_true", help="skip the extra one-step-at-a-time leg")
    ap.add_argument("--seed", type=int, default=20250711)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the result: everything else that writes to file descriptor 1 during the run (RCCL prints a
    # version banner there when a communicator is created) is sent to stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    shared_gpu = world > n_dev              # debug aid: several ranks on one GPU (RCCL refuses that: gloo callbacks instead)
    local = local % max(1, n_dev)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    wl = args.workload if args.workload != "auto" else ("c3" if world == 1 else "c4")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SYLPH_BENCH_BACKEND", "gloo" if shared_gpu else "nccl")   # nccl == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    c, k, read_len = 200, 31, 150
    n_pairs, _, _, _, _, spg, n_sets = WORKLOADS[wl]
    if args.samples_per_step:
        spg = args.samples_per_step
        n_sets = max(n_sets, spg)
    long_mode = wl == "c5"
    c_reads = 100 if long_mode else c        # reads may be sketched denser than the DB (contain.rs:562-568,616-623)
    # One HIP stream for everything: the library launches on a torch-owned stream, so torch-side generation, the
    # library's kernels and the timing events are stream-ordered without cross-queue synchronisation.
    tstream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(tstream)
    ctx = S.Context(local, stream=tstream.cuda_stream)
    for kv in filter(None, os.environ.get("SYLPH_BENCH_CTX_OPTIONS", "").split(",")):   # tuning experiments only
        ctx.set_option(*kv.split("=", 1))

    db_mode = args.db_mode if args.db_mode != "auto" else ("shard" if world > 1 else "replicate")
    log(f"[bench] building workload {wl} on {world} GPU(s), db {db_mode} ...")
    db, n_total, community, dbstats, verify_set = build_database(ctx, device, wl, c, k, args.seed, rank, world, db_mode)
    comm, comm_kind, fallbacks = None, None, []

    def agreed_failure(failed):            # a rank that failed takes every rank down the same fallback
        if dist is None:
            return bool(failed)
        t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item())

    if db_mode == "shard":
        if world == 1:      # the sharded code path with a one-rank RCCL communicator: the exchange's own cost, nothing on the wire
            comm, comm_kind = S.Comm(0, 1, ctx=ctx, rccl_id=S.Comm.rccl_unique_id()), "RCCL (one rank)"
        elif shared_gpu or dist.get_backend() != "nccl":
            comm, comm_kind = SH.torch_callback_comm(dist, device), "torch.distributed callbacks on host copies, ranks share a GPU"
        else:
            err = None
            if os.environ.get("SYLPH_BENCH_COMM", "library") == "library":
                try:
                    comm, comm_kind = SH.rccl_comm(dist, ctx, device), "RCCL, communicator created by the library"
                except Exception as e:
                    err = e
            else:
                err = "SYLPH_BENCH_COMM"
            if agreed_failure(err is not None):
                if comm is not None:
                    comm.close()
                fallbacks.append(f"library RCCL communicator: {err}")
                comm, comm_kind = SH.torch_device_comm(dist, device), "RCCL through torch.distributed collectives on the device buffers"
    t0 = time.time()
    read_sets = []                           # distinct samples, rotated over the steps so that the probe is never cache-warm
    for i in range(n_sets):
        sd = args.seed + 1_000_003 * (rank + 1) + 7919 * i
        if long_mode:
            bases, rec_off = synth.long_reads(community, 5_000_000_000, seed=sd)
            n_records = rec_off.numel() - 1
            cuts = [0]                       # a push holds < 2^32 bases: batches of whole reads
            while cuts[-1] < n_records:
                nxt = int(torch.searchsorted(rec_off, rec_off[cuts[-1]] + 3_000_000_000).item())
                cuts.append(max(cuts[-1] + 1, min(nxt, n_records)))
            batches = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                o = (rec_off[a:b + 1] - rec_off[a]).contiguous()
                batches.append((int(rec_off[a].item()), o, b - a, int(o[-1].item())))
            read_sets.append(dict(bases=bases, rec_off=rec_off, n_bases=int(rec_off[-1].item()), n_records=n_records, batches=batches))
        else:
            if wl == "c3r":
                bases, rec_off = synth.ragged_paired_reads(community, n_pairs, seed=sd)
            else:
                bases, rec_off = synth.paired_reads(community, n_pairs, read_len=read_len, seed=sd)
            read_sets.append(dict(bases=bases, rec_off=rec_off, n_bases=int(rec_off[-1].item()), n_records=2 * n_pairs, batches=None))
    torch.cuda.synchronize()
    del community
    n_bases = float(np.mean([r["n_bases"] for r in read_sets]))
    log(f"[bench] db {dbstats}; {n_sets} read sets of {n_bases / 1e9:.3f} Gbp generated in {time.time() - t0:.1f}s")

    last = {}
    # Sketching runs on worker threads, each with its own context (stream, memory pool); the profile stage runs on the main
    # thread's context.  --pipeline-depth steps are in flight at once: while step i is profiled, the samples of the next steps
    # are already being sketched, and the small launch-bound kernels of one sample's dedup/count stage fill the gaps beside
    # another sample's seeding kernel.  (The reference sketches samples on parallel threads too, sketch.rs:313, and profiles
    # sample after sample against the loaded database, contain.rs:267-289.)  Depth 1 = one step at a time.
    import queue
    import threading
    from collections import deque
    from concurrent.futures import Future
    n_workers = max(1, args.sketch_workers or 2)
    depth = max(1, args.pipeline_depth or (n_workers + 1 if spg == 1 else 2))

    class SketchWorker(threading.Thread):
        def __init__(self):
            super().__init__(daemon=True)
            self.ctx = S.Context(local)
            self.jobs = queue.Queue()
            self.start()

        def run(self):
            torch.cuda.set_device(local)
            while True:
                job = self.jobs.get()
                if job is None:
                    return
                fn, fut = job
                try:
                    fut.set_result(fn(self.ctx))
                except BaseException as e:       # handed to whoever waits for the result
                    fut.set_exception(e)

        def submit(self, fn):
            fut = Future()
            self.jobs.put((fn, fut))
            return fut

    workers = [SketchWorker() for _ in range(n_workers)]
    all_ctx = [ctx] + [w.ctx for w in workers]
    for w in workers:
        for kv in filter(None, os.environ.get("SYLPH_BENCH_CTX_OPTIONS", "").split(",")):
            w.ctx.set_option(*kv.split("=", 1))
    sample_no = [0]

    def sketch_one(wctx, rs):
        t_a = time.perf_counter()
        sk = S.ReadSketcher(wctx, c=c_reads, k=k, paired=not long_mode)
        if long_mode:
            for start, o, nrec, nb in rs["batches"]:
                sk.push_device(rs["bases"].data_ptr() + start, o.data_ptr(), nrec, nb)
        else:
            sk.push_device(rs["bases"].data_ptr(), rs["rec_off"].data_ptr(), rs["n_records"], rs["n_bases"])
        dk, dc, n, dup = sk.finish_device()
        return sk, (dk, dc, n, dup), time.perf_counter() - t_a

    class Inline:                    # one step at a time with one sample per step: everything on the main thread's context
        @staticmethod
        def submit(fn):
            fut = Future()
            fut.set_result(fn(ctx))
            return fut

    def submit_step(inline=False):
        jobs = []
        for _ in range(spg):
            rs = read_sets[sample_no[0] % n_sets]
            w = Inline if inline else workers[sample_no[0] % n_workers]
            jobs.append((w, w.submit(lambda wctx, rs=rs: sketch_one(wctx, rs))))
            sample_no[0] += 1
        return jobs

    def finish_step(jobs, collect=None):
        done = [f.result() for _, f in jobs]
        tables = [d[1] for d in done]
        t_b = time.perf_counter()
        refs = [(dk, dc, n) for dk, dc, n, _ in tables]
        if comm is not None:
            res = db.contain_batch_sharded(comm, refs, device_ptrs=True)
        else:
            res = db.contain_batch(refs, device_ptrs=True)      # borrowed pinned views
        t_c = time.perf_counter()
        if collect == "final":       # untimed extra step: keep what the verify / roofline legs need
            last["res"] = tuple(np.array(x) for x in res)
            last["table"] = tables[0][:3]
            last["occ"] = [int(SH.device_view(dc, n, torch.int32, device).sum().item()) + dup for dk, dc, n, dup in tables]
            last["n_table"] = [n for _, _, n, _ in tables]
            last["hits"] = int(len(res[2]))
            last["sessions"] = [(w, d[0]) for (w, _), d in zip(jobs, done)]
            return
        for (w, _), d in zip(jobs, done):        # a session is closed on the thread that owns its context
            w.submit(lambda wctx, sk=d[0]: sk.close())
        if isinstance(collect, list):
            collect.append((float(np.sum([d[2] for d in done])), t_c - t_b, [t[2] for t in tables], [t[3] for t in tables], len(res[2])))

    def run_steps(n, collect=None, in_flight=1):
        pending = deque()
        submitted = 0
        for i in range(n):
            while submitted < n and submitted < i + in_flight:      # steps i .. i + in_flight - 1 are in flight while i is finished
                pending.append(submit_step(inline=(in_flight == 1 and spg == 1)))
                submitted += 1
            finish_step(pending.popleft(), collect)
        for w in workers:                                            # the sessions' close jobs
            w.submit(lambda wctx: None).result()

    # untimed settle steps (first-use allocations of every context's pool, lazy kernel loading) whatever --warmup is
    if comm is not None and world > 1:
        # the first sharded step ever run on this node: if the exchange fails on any rank, every rank drops to the replicated
        # database (no data-path collective) and says so in the result line
        err = None
        try:
            run_steps(1, None, 1)
        except Exception as e:
            err = e
        if agreed_failure(err is not None):
            fallbacks.append(f"sharded exchange ({comm_kind}): {err}")
            log(f"[bench] rank {rank}: sharded step failed ({err}); falling back to the replicated database")
            try:
                comm.close()
                db.close()
            except Exception:
                pass
            comm, comm_kind, db_mode = None, None, "replicate"
            db, n_total, _, dbstats, verify_set = build_database(ctx, device, wl, c, k, args.seed, rank, world, db_mode)
    run_steps(2 * n_workers, None, 2)
    run_steps(2, None, 1)
    run_steps(2 * depth, None, depth)      # ... and with as many sessions alive per context as the pipelined region will have
    torch.cuda.synchronize()
    run_steps(args.warmup, None, depth)

    def timed(n, in_flight):
        for wc in all_ctx:
            wc.profile(not args.no_kernel_timers)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        rows = []
        run_steps(n, rows, in_flight)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        fam = {}
        for f in ("seeds", "compact", "annotate", "sort", "replay", "probe", "exchange"):   # HIP events on the launch streams, all contexts
            tot = [0.0, 0]
            for wc in all_ctx:
                ms, nl = wc.kernel_stats(f)
                tot[0] += ms
                tot[1] += nl
            fam[f] = tuple(tot)
        for wc in all_ctx:
            wc.profile(False)
        return elapsed, rows, fam

    elapsed, rows, fam = timed(args.steps, depth)
    one_at_a_time = None
    if depth > 1 and not args.no_sequential_leg:     # the same steps one at a time: step latency, kernels measured alone on the GPU
        e1, r1, f1 = timed(min(args.steps, 8), 1)
        one_at_a_time = (e1 / min(args.steps, 8), r1, f1)

    finish_step(submit_step(), "final")   # untimed: seed occurrences for the roofline's algorithmic bytes, results for --verify
    t_sketch = float(np.mean([r[0] for r in rows])) / min(n_workers, spg * depth)   # busy time of the sketch workers per step
    t_profile = float(np.mean([r[1] for r in rows]))
    ms_per_step = elapsed / args.steps * 1e3
    value = world * spg * n_bases / 1e9 / (elapsed / args.steps)              # whole-job read Gbp/s through both stages
    comparisons = world * spg * n_total                                       # every sample vs every genome of the DB
    parallelism = (f"{spg} sample(s) per GPU per step x {world} GPU(s); database " +
                   (f"sharded by k-mer range over {world} GPUs (per step: table slices all-to-all, hits all-to-all to the owners, two tiny all-gathers of sizes; "
                    f"{comm_kind})"
                    if comm is not None else ("replicated on every GPU (no data-path collective)" if world > 1 else "on the one GPU")))
    out = {
        "metric": "read Gbp/s sketched + genome-comparisons/s profiled, 1 Gbp vs GTDB-R220",
        "value": round(value, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": DESCR[wl], "samples_per_gpu_per_step": spg, "sketch_workers_per_gpu": n_workers, "steps_in_flight": depth, **({"fallbacks": fallbacks} if fallbacks else {}), "reads_per_sample_gbp": round(n_bases / 1e9, 4),
                   "distinct_read_sets_rotated": n_sets, "genomes": n_total, "db_kmers_per_shard": dbstats["shard_kmers"],
                   "dedup": "exact (--fpr 0 semantics)" if not long_mode else "none applies (reads > 400 bp, sketch.rs:922-927)",
                   "seed_mode": "avx2_compat", "parallelism": parallelism,
                   "inputs": "reads + database resident in HBM before the timed region",
                   "rng": "torch (Philox) generators on the device, seed 20250711 + 1000003*(rank+1) + 7919*set — not the splitmix64 streams of SURVEY 8d"},
        "sketch_gbp_per_s": round(world * spg * n_bases / 1e9 / t_sketch, 3),
        "genome_comparisons_per_s": round(comparisons / t_profile, 1),
        "sketch_ms": round(t_sketch * 1e3, 3), "profile_ms": round(t_profile * 1e3, 3),
        "sample_table_entries": int(np.mean([np.mean(r[2]) for r in rows])), "dup_removed": int(np.mean([np.mean(r[3]) for r in rows])),
        "kernel_ms": {f: (round(v[0] / max(1, v[1]), 4), int(v[1])) for f, v in fam.items() if v[1]},
        "setup": dbstats,
    }
    # roofline of the dominant kernel (seeds): algorithmic bytes per launch = 1 B/base + 8 B/record offset + 8 B/seed
    # occurrence out, over the HIP-event duration of that launch.
    seeds_ms, seeds_launches = fam["seeds"]
    if seeds_launches:
        n_rec = float(np.mean([r["n_records"] for r in read_sets]))
        n_occ = float(np.mean(last["occ"]))
        launches_per_sample = max(1, round(seeds_launches / (args.steps * spg)))
        alg_bytes = (n_bases + 8 * n_rec + 8 * n_occ) / launches_per_sample   # a sample > 2^32 bases is pushed in batches
        avg_ms = seeds_ms / seeds_launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "seeds_traffic.json")
        if os.path.exists(tf) and wl in ("c2", "c3", "c4"):   # measured on this read set with rocprofv3 --pmc (profiles/)
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        meta = {}
        try:
            meta = json.load(open(os.path.join(ROOT, "profiles", "seeds_traffic.json")))
        except Exception:
            pass
        ipk = meta.get("valu_per_kmer_position_kernel", 38) if long_mode else meta.get("valu_per_kmer", 44)
        hashed = (n_bases if long_mode else max(0.0, n_bases - n_rec * (k - 1))) / launches_per_sample
        out["roofline"] = {"bound": "hbm", "kernel": "seeds_slots_kernel<31,1>" if long_mode else "reads_kernel<31,1>", "achieved": round(achieved, 1),
                           "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                           "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(avg_ms, 4),
                           "note": f"integer-VALU issue bound: {ipk} VALU wave-instructions per hashed k-mer (SQ counters in profiles/)" +
                                   (f"; {depth} steps in flight on {n_workers} sketch streams + the profile stream: launch durations include time shared with the other streams' kernels (alone on the GPU: one_step_at_a_time)" if depth > 1 else ""),
                           # secondary ceiling: VALU issue = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz lane-ops/s, i.e. 4 cycles per
                           # wave-instruction; plain VOP2 integer ops issue faster than that on this chip (profiles/r02_valu_rates.txt),
                           # so a launch alone on the GPU can come out a few per cent above 1.0
                           "valu_ceiling": {"instr_per_kmer": ipk, "kmers_per_launch": int(hashed),
                                            "min_ms": round(ipk * hashed / (256 * 4 * 16 * 2.4e9) * 1e3, 4),
                                            "frac": round(ipk * hashed / (256 * 4 * 16 * 2.4e9) * 1e3 / avg_ms, 3)}}
    # roofline of the profile half: probe_kernel, one launch per step over all tables it probes.  Inverted-index formulation
    # (SURVEY 8d): B = N_s * (8 + 4 table in + 64 one index line per probe) + 8 * hits out.
    probe_ms, probe_launches = fam["probe"]
    if probe_launches:
        probes = float(np.mean([np.sum(r[2]) for r in rows]))      # table entries of this rank's samples per step ...
        hits = float(np.mean([r[4] for r in rows]))
        # ... which, sharded, is also what this rank probes: 1/N of each of the N x spg tables of the step
        alg = probes * (12 + 64) + 8 * hits
        avg = probe_ms / probe_launches
        ptraffic = None
        try:       # FETCH_SIZE + WRITE_SIZE per probe of the C3 probe (rocprofv3 --pmc passes, profiles/r02_kernel_stats.md) x this launch's probes
            per_probe = json.load(open(os.path.join(ROOT, "profiles", "seeds_traffic.json"))).get("probe_hbm_bytes_per_probe")
            if per_probe and wl in ("c3", "c4", "c3r"):
                ptraffic = int(per_probe * probes)
        except Exception:
            ptraffic = None
        out["roofline_profile"] = {"bound": "hbm", "kernel": "probe_kernel", "achieved": round(alg / (avg * 1e-3) / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(alg / (avg * 1e-3) / 1e9 / 8000.0, 4), "traffic": ptraffic,
                                   "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(avg, 4),
                                   "probes_per_launch": int(probes), "hits_per_launch": int(hits),
                                   "note": "random 64 B line reads, latency-bound; one index line per probe is the access granule"}
    if one_at_a_time is not None:
        s1, r1, f1 = one_at_a_time
        o1 = {"ms_per_step": round(s1 * 1e3, 3), "value": round(world * spg * n_bases / 1e9 / s1, 3),
              "sketch_ms": round(float(np.mean([r[0] for r in r1])) / min(n_workers, spg) * 1e3, 3), "profile_ms": round(float(np.mean([r[1] for r in r1])) * 1e3, 3),
              "kernel_ms": {f: (round(v[0] / max(1, v[1]), 4), int(v[1])) for f, v in f1.items() if v[1]}}
        if "roofline" in out and f1["seeds"][1]:
            a1 = f1["seeds"][0] / f1["seeds"][1]
            o1["roofline_frac"] = round(out["roofline"]["algorithmic_bytes_per_launch"] / (a1 * 1e-3) / 1e9 / 8000.0, 4)
            o1["valu_ceiling_frac"] = round(out["roofline"]["valu_ceiling"]["min_ms"] / a1, 3)
            # the same launches with nothing else on the GPU (same run, HIP events of the one-step-at-a-time leg): the kernel's own speed
            out["roofline"]["alone_on_gpu"] = {"avg_launch_ms": round(a1, 4), "achieved": round(out["roofline"]["algorithmic_bytes_per_launch"] / (a1 * 1e-3) / 1e9, 1),
                                               "frac": o1["roofline_frac"], "launches": int(f1["seeds"][1])}
        if "roofline_profile" in out and f1["probe"][1]:
            p1 = f1["probe"][0] / f1["probe"][1]
            o1["roofline_profile_frac"] = round(out["roofline_profile"]["algorithmic_bytes_per_launch"] / (p1 * 1e-3) / 1e9 / 8000.0, 4)
            out["roofline_profile"]["alone_on_gpu"] = {"avg_launch_ms": round(p1, 4), "achieved": round(out["roofline_profile"]["algorithmic_bytes_per_launch"] / (p1 * 1e-3) / 1e9, 1),
                                                       "frac": o1["roofline_profile_frac"], "launches": int(f1["probe"][1])}
        out["one_step_at_a_time"] = o1
    if not args.no_verify and last.get("res") is not None:
        try:
            G = n_total
            cc, off, covs = last["res"]
            out["verify"] = verify_against_oracle(ctx, (cc[:G], off[:G + 1], covs), G, last["table"], verify_set, device)
        except Exception as e:
            out["verify"] = {"genomes_checked": 0, "mismatches": None, "error": str(e)}
    for w, sk in last.get("sessions", []):
        w.submit(lambda wctx, sk=sk: sk.close()).result()
    # ---- host-fed leg (untimed w.r.t. `value`): the same step with the reads starting in PAGE-LOCKED HOST memory, as ASCII and
    # as the packed 2-bit stream a feed would hand over (sylph_sketch_push_enc cuts the batch into chunks that travel on a copy
    # stream while the previous chunk is sketched): pinned host -> HBM -> sketch -> profile -> results on the host.
    if rank == 0 and world == 1 and not args.no_h2d and not long_mode:
        try:
            from sylph_amd.binding import ENC_2BIT, ENC_ASCII, MEM_HOST_PINNED
            rs = read_sets[0]
            nb, nrec = rs["n_bases"], rs["n_records"]
            host_ascii = rs["bases"][:nb].cpu().numpy()
            off_h = rs["rec_off"].cpu().numpy().astype(np.uint64)
            packed = S.pack_2bit(host_ascii)
            pin_a, pin_p, pin_o = S.PinnedBuffer(nb + 64), S.PinnedBuffer(len(packed) + 64), S.PinnedBuffer(len(off_h) * 8)
            pin_a.array[:nb] = host_ascii
            pin_p.array[:len(packed)] = packed
            pin_o.array.view(np.uint64)[:len(off_h)] = off_h
            del host_ascii, packed
            # the link itself: one page-locked 1 GiB host -> device copy
            probe_t = torch.empty(1 << 30, dtype=torch.uint8, device=device)
            src_t = torch.empty(1 << 30, dtype=torch.uint8, pin_memory=True)
            bw = []
            for _ in range(3):
                torch.cuda.synchronize()
                tq = time.perf_counter()
                probe_t.copy_(src_t, non_blocking=True)
                torch.cuda.synchronize()
                bw.append((1 << 30) / (time.perf_counter() - tq) / 1e9)
            del probe_t, src_t
            link = max(bw)
            h2d = {"pinned_h2d_gb_per_s": round(link, 1)}
            for name, enc, pin, nbytes in (("ascii", ENC_ASCII, pin_a, nb), ("2bit", ENC_2BIT, pin_p, (nb + 3) // 4)):
                ts = []
                for rep in range(4):
                    torch.cuda.synchronize()
                    tq = time.perf_counter()
                    sk = S.ReadSketcher(ctx, c=c_reads, k=k, paired=True)
                    sk.push_enc(pin.ptr, pin_o.ptr, nb, MEM_HOST_PINNED, enc, n_records=nrec)
                    dk, dc, n, dup = sk.finish_device()
                    res = db.contain_batch([(dk, dc, n)], device_ptrs=True)
                    ts.append(time.perf_counter() - tq)
                    sk.close()
                t_best = float(np.median(ts[1:]))
                moved = nbytes + 8 * (nrec + 1)
                h2d[name] = {"gbp_per_s": round(nb / 1e9 / t_best, 2), "ms_per_step": round(t_best * 1e3, 3), "bytes_over_pcie": int(moved),
                             "pcie_frac": round(moved / t_best / 1e9 / link, 3)}
            out["value_h2d_inclusive"] = h2d
            for p_ in (pin_a, pin_p, pin_o):
                p_.close()
        except Exception as e:
            out["value_h2d_inclusive"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not long_mode and wl != "c3r":
        try:
            from oracle import oracle as O  # noqa: F401  (cpu_baseline leg only)
            G_s = min(db.n_genomes, 16000)
            # bounded DB sample for the CPU probe: the oracle needs the genome-major layout, regenerate decoys of that size
            dk, doff = synth.decoy_sketches(G_s, c=c, device=device, seed=args.seed + 7)
            rs = read_sets[0]
            cb = cpu_baseline(rs["bases"], rs["rec_off"], n_pairs, read_len, c, k, (dk.cpu().numpy().view(np.uint64), doff.cpu().numpy().astype(np.uint64)))
            t_cpu = n_bases / 1e9 / cb["sketch_gbp_per_s"] + n_total / cb["comparisons_per_s"]
            out["cpu_baseline"] = {"value": round(n_bases / 1e9 / t_cpu, 4), "unit": "Gbp/s", "cores": cb["probe_cores"], "kind": "port",
                                   "sample": cb["sample"], "sketch_gbp_per_s": round(cb["sketch_gbp_per_s"], 4),
                                   "sketch_cores": 1, "genome_comparisons_per_s": round(cb["comparisons_per_s"], 1),
                                   "note": "C++ restatement of the reference CPU path (oracle/); the reference sketches one sample on one thread and probes genomes on all threads; value = extrapolation to one whole step"}
        except Exception as e:  # the baseline leg must never sink the GPU measurement
            out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    sys.stdout.flush()
    try:                      # C stdio of the libraries (RCCL's banner) is still buffered: flush it while fd 1 is stderr
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)
    db.close()
    if comm is not None:
        comm.close()
    for w in workers:
        w.jobs.put(None)
        w.join()
        w.ctx.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
