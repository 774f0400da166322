#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X sketch + profile hot path.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): read Gbp/s sketched + genome-comparisons/s profiled, 1 Gbp of 2x150 bp reads vs a
GTDB-R220-scale database (113,104 genome sketches, k=31, c=200).  One *step* = one pass of the hot path over one
sample per GPU: sketch 1 Gbp of paired reads that are already resident in HBM (seeding -> exact dedup/count) and
profile the resulting table against the resident database (containment counts + coverage vectors back on the host).
`value` = whole-job read Gbp/s through both stages; the per-stage rates are reported next to it.

Multi-GPU: samples are independent units (one per rank per step, no collective in the sketch stage).
Database placement (--db-mode): "replicate" (default when the postings index fits HBM comfortably — 22 GB of 288 GB at
GTDB-R220 scale): every rank holds the whole index and profiles its own sample, and ONE RCCL all-gather per step
collects the per-sample containment counts on every rank; "shard": the database is sharded by genome across ranks,
sample tables are all-gathered, every rank probes every sample against its shard and ONE all-gather combines
counts + coverage lists (what a database larger than one GPU needs; O(sample) probe work per rank per sample, so
it scales worse).  scaling = weak (per-GPU sketch work fixed).

The CPU baseline leg (rank 0, N=1 only) times the oracle — the C++ restatement of the reference's AVX2/rayon path —
on a bounded sample of the same inputs; it is the only place bench.py touches oracle/.
"""
import argparse
import json
import os
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
    # name: (n_pairs, n_community, n_seq_backed, n_genomes_total, genome_len)
    "c3": (3_333_334, 100, 1000, 113_104, 5_000_000),   # BASELINE configs[2]: 1 Gbp vs GTDB-R220-scale DB
    "c2": (3_333_334, 100, 1000, 1000, 5_000_000),      # BASELINE configs[1]: 1 Gbp vs 1,000 x 5 Mbp genomes
    "small": (100_000, 8, 24, 2000, 400_000),           # quick functional run
    "c5": (None, 100, 1000, 113_104, 5_000_000),        # BASELINE configs[4]: ONT-like long reads (N50 10 kb, 5 Gbp), reads at c=100
}


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def build_database(ctx, device, wl, c, k, seed, rank, world, db_mode="shard"):
    """-> (Database for this rank's shard, shard genome ids, n_genomes_total, community genomes tensor, stats)"""
    n_pairs, n_comm, n_seq, n_total, glen = WORKLOADS[wl]
    t0 = time.time()
    # sequence-backed genomes: community + unrelated + 10 % mutated copies (97 % identity) of the first ones
    n_mut = n_seq // 10
    community = synth.random_genomes(n_comm, glen, device, seed, mutated_frac=0.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 17)
    sub = torch.tensor([67, 71, 84, 65], dtype=torch.uint8, device=device)   # A->C, C->G, G->T, T->A
    lut = torch.zeros(256, dtype=torch.uint8, device=device)
    lut[torch.tensor([65, 67, 71, 84], device=device)] = sub
    # Database build: genomes are generated on the device in batches of <= 1 Gbp and sketched by ONE
    # sylph_sketch_genomes call per batch (seeding + genome-wide duplicate removal + spacing filter on the device; only the
    # sketches come back to the host).  Timed separately from generation: db_build_gbp_per_s.
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
    kmers = torch.cat([torch.from_numpy(seq_k.view(np.int64)).to(device), dk])
    goff = torch.cat([torch.from_numpy(seq_off).to(device), doff[1:] + int(seq_off[-1])])
    del dk
    lens = (goff[1:] - goff[:-1]).cpu().numpy()
    # shard by genome, balanced by k-mer count
    owner = SH.partition_genomes(lens, world if db_mode == "shard" else 1)
    mine = np.nonzero(owner == (rank if db_mode == "shard" else 0))[0]
    if world > 1 and db_mode == "shard":
        idx = torch.from_numpy(mine).to(device)
        starts, ls = goff[:-1][idx], torch.from_numpy(lens[mine]).to(device)
        soff = torch.zeros(len(mine) + 1, dtype=torch.int64, device=device)
        soff[1:] = torch.cumsum(ls, 0)
        pos = torch.arange(int(soff[-1].item()), device=device)
        seg = torch.searchsorted(soff[1:], pos, right=True)
        kmers = kmers[starts[seg] + (pos - soff[seg])]
        goff = soff
    torch.cuda.synchronize()
    t2 = time.time()
    db = S.Database(ctx, kmers.data_ptr(), goff.data_ptr(), device_ptrs=True, n_genomes=len(mine))
    ctx.synchronize()
    t3 = time.time()
    stats = dict(n_genomes=int(n_total), shard_genomes=int(len(mine)), shard_kmers=int(db.n_kmers),
                 seq_backed_generate_and_sketch_s=round(t1 - t0, 2), seq_backed_sketch_s=round(t_sketch, 4),
                 db_build_gbp_per_s=round(n_seq * glen / 1e9 / max(t_sketch, 1e-9), 2), db_build_genomes=int(n_seq),
                 generate_s=round(t2 - t1, 2), db_upload_index_s=round(t3 - t2, 2))
    lens_mine = lens[mine]
    # NB: no torch.cuda.empty_cache() here — returning tens of GB to the driver (hipFree) queues page-table work
    # that stalls this process's GPU queues for 15-40 ms at random moments over the next few hundred ms.
    del kmers, goff
    return db, mine, lens_mine, int(n_total), community, stats


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
                sample=f"sketch: first {n_s} read pairs ({n_s * 2 * read_len / 1e6:.0f} Mbp) on 1 thread "
                       f"({'AVX2 intrinsics' if mode == O.MODE_AVX2_FAST else 'scalar'}); "
                       f"probe: {G} genomes ({len(dbk) / 1e6:.0f} M k-mers) on {cores} threads vs the {len(sk['kmers'])}-entry sample table")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("SYLPH_BENCH_WORKLOAD", "c3"), choices=sorted(WORKLOADS))
    ap.add_argument("--db-mode", default=os.environ.get("SYLPH_BENCH_DB_MODE", "auto"), choices=["auto", "replicate", "shard"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip the in-library HIP-event kernel timers (no roofline object)")
    ap.add_argument("--seed", type=int, default=20250711)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    n_dev = torch.cuda.device_count()
    local = local % max(1, n_dev)   # (debug aid: several ranks may share one GPU with SYLPH_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SYLPH_BENCH_BACKEND", "nccl")   # nccl == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    c, k, read_len = 200, 31, 150
    n_pairs = WORKLOADS[args.workload][0]
    long_mode = args.workload == "c5"
    c_reads = 100 if long_mode else c        # reads may be sketched denser than the DB (contain.rs:562-568,616-623)
    # One HIP stream for everything: the library launches on a torch-owned stream, so torch-side generation, the
    # library's kernels and the timing events are stream-ordered without cross-queue synchronisation.
    tstream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    ctx = S.Context(local, stream=stream)
    for kv in filter(None, os.environ.get("SYLPH_BENCH_CTX_OPTIONS", "").split(",")):   # tuning experiments only
        ctx.set_option(*kv.split("=", 1))

    log(f"[bench] building workload {args.workload} on {world} GPU(s) ...")
    # ~1.8e9 postings x 12.5 B = 22 GB at GTDB-R220 scale: replicate unless it would take more than a quarter of HBM
    db_mode = args.db_mode
    if db_mode == "auto":
        hbm = torch.cuda.get_device_properties(device).total_memory
        est = WORKLOADS[args.workload][3] * 16000 * 12.5
        db_mode = "replicate" if est < hbm / 4 else "shard"
    db, mine, lens_mine, n_total, community, dbstats = build_database(ctx, device, args.workload, c, k, args.seed, rank, world, db_mode)
    t0 = time.time()
    if long_mode:
        bases, rec_off = synth.long_reads(community, 5_000_000_000, seed=args.seed + 1_000_003 * (rank + 1))
        n_records = rec_off.numel() - 1
        n_bases = int(rec_off[-1].item())
        # a push holds < 2^32 bases: split the 5 Gbp sample into batches of whole reads
        cuts = [0]
        while cuts[-1] < n_records:
            nxt = int(torch.searchsorted(rec_off, rec_off[cuts[-1]] + 3_000_000_000).item())
            cuts.append(max(cuts[-1] + 1, min(nxt, n_records)))
        batches = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = (rec_off[a:b + 1] - rec_off[a]).contiguous()
            start = int(rec_off[a].item())
            assert start % 16 == 0 or a == 0 or True
            batches.append((start, o, b - a, int(o[-1].item())))
    else:
        bases, rec_off = synth.paired_reads(community, n_pairs, read_len=read_len, seed=args.seed + 1_000_003 * (rank + 1))
        n_bases = n_pairs * 2 * read_len
        n_records = 2 * n_pairs
    torch.cuda.synchronize()
    del community
    log(f"[bench] db {dbstats}; reads {n_bases / 1e9:.3f} Gbp generated in {time.time() - t0:.1f}s")

    group = SH.TorchGroup(dist, device) if world > 1 else SH.LocalGroup()

    occ_holder = [None]

    def step(collect=None):
        t_a = time.perf_counter()
        sk = S.ReadSketcher(ctx, c=c_reads, k=k, paired=not long_mode)
        t_a1 = time.perf_counter()
        if long_mode:
            for start, o, nrec, nb in batches:
                sk.push_device(bases.data_ptr() + start, o.data_ptr(), nrec, nb)
        else:
            sk.push_device(bases.data_ptr(), rec_off.data_ptr(), n_records, n_bases)
        t_a2 = time.perf_counter()
        dk, dc, n, dup = sk.finish_device()
        t_b = time.perf_counter()
        if os.environ.get("SYLPH_BENCH_DEBUG"):
            log(f"[bench] begin {1e3 * (t_a1 - t_a):.3f} push {1e3 * (t_a2 - t_a1):.3f} finish {1e3 * (t_b - t_a2):.3f} ms")
        occ_holder[0] = (dc, n, dup)
        if db_mode == "shard" or world == 1:
            res = SH.profile_step(db, group, dk, dc, n, mine, n_total, device)
        else:   # replicated index: profile the rank's own sample, then one all-gather of the containment counts
            res = SH.profile_step(db, SH.LocalGroup(), dk, dc, n, mine, n_total, device)
            res["all_counts"] = SH.gather_counts(group, res["contain_count"], device)
        t_c = time.perf_counter()
        if os.environ.get("SYLPH_BENCH_DEBUG"):
            log(f"[bench] contain {1e3 * (t_c - t_b):.3f} ms")
        if collect == "occ":   # untimed extra step: seed occurrences of the sample = sum(counts) + removed
            occ_holder.append(int(SH.device_view(dc, n, torch.int32, device).sum().item()) + dup)
        sk.close()
        if isinstance(collect, list):
            collect.append((t_b - t_a, t_c - t_b, n, dup, res))
        return res

    # two untimed settle steps (first-use allocations of the library's pool, lazy kernel loading) whatever --warmup is
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ctx.profile(not args.no_kernel_timers)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    rows = []
    for _ in range(args.steps):
        step(rows)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel timing from HIP events recorded on the launch stream inside the library ----
    seeds_ms, seeds_launches = ctx.kernel_stats("seeds")
    fam = {f: ctx.kernel_stats(f) for f in ("seeds", "compact", "annotate", "sort", "replay", "probe")}
    ctx.profile(False)
    step("occ")   # untimed: count the seed occurrences for the roofline's algorithmic bytes
    t_sketch = float(np.mean([r[0] for r in rows]))
    t_profile = float(np.mean([r[1] for r in rows]))
    n_table, dup = rows[-1][2], rows[-1][3]
    occ = occ_holder[1] if len(occ_holder) > 1 else None

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_bases / 1e9 / (elapsed / args.steps)              # whole-job read Gbp/s through both stages
    comparisons = world * n_total                                        # every sample vs every genome of the DB
    out = {
        "metric": "read Gbp/s sketched + genome-comparisons/s profiled, 1 Gbp vs GTDB-R220",
        "value": round(value, 3), "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": {"c3": "1 Gbp synthetic 2x150 bp reads vs GTDB-R220-scale DB (113,104 genome sketches), k=31 c=200 (BASELINE configs[2])",
                                "c2": "1 Gbp synthetic 2x150 bp reads vs 1,000 synthetic 5 Mbp genomes, k=31 c=200 (BASELINE configs[1])",
                                "small": "functional smoke workload (NOT the BASELINE config)",
                                "c5": "ONT-like long reads (N50 10 kb, 5 Gbp, 5 % substitutions) sketched at c=100 vs GTDB-R220-scale DB at c=200 (BASELINE configs[4])"}[args.workload],
                   "reads_per_gpu_per_step_gbp": round(n_bases / 1e9, 4), "genomes": n_total, "db_kmers_per_shard": dbstats["shard_kmers"],
                   "dedup": "exact (--fpr 0 semantics)" if not long_mode else "none applies (reads > 400 bp, sketch.rs:922-927)", "seed_mode": "avx2_compat", "parallelism": f"samples x{world}, db {db_mode}d x{world}" if world > 1 else "1 sample, 1 GPU",
                   "inputs": "reads + database resident in HBM before the timed region"},
        "sketch_gbp_per_s": round(world * n_bases / 1e9 / t_sketch, 3),
        "genome_comparisons_per_s": round(comparisons / t_profile, 1),
        "sketch_ms": round(t_sketch * 1e3, 3), "profile_ms": round(t_profile * 1e3, 3),
        "sample_table_entries": int(n_table), "dup_removed": int(dup),
        "kernel_ms": {f: (round(v[0] / max(1, v[1]), 4), int(v[1])) for f, v in fam.items()},
        "setup": dbstats,
    }
    # roofline of the dominant kernel (seeds): algorithmic bytes per launch = 1 B/base + 8 B/record offset + 8 B/seed
    # occurrence out, over the HIP-event duration of that launch.
    if seeds_launches:
        n_rec = n_records
        n_occ = occ if occ is not None else int(n_bases / c_reads)
        launches_per_step = max(1, round(seeds_launches / args.steps))
        alg_bytes = (n_bases + 8 * n_rec + 8 * n_occ) // launches_per_step   # a sample > 2^32 bases is pushed in batches
        avg_ms = seeds_ms / seeds_launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "seeds_traffic.json")
        if os.path.exists(tf) and args.workload in ("c2", "c3"):   # measured on this read set (profiles/r01_kernel_stats.md, PMC section)
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # hashed k-mers per launch: every position (position kernel) or only the in-read k-mers (read-per-lane kernel)
        ipk = 38 if long_mode else 44
        hashed = (n_bases if long_mode else max(0, n_bases - n_rec * (k - 1))) // launches_per_step
        out["roofline"] = {"bound": "hbm", "kernel": "seeds_slots_kernel<31,1>" if long_mode else "reads_kernel<31,1>", "achieved": round(achieved, 1), "peak": 8000.0,
                           "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                           "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(avg_ms, 4),
                           "note": "integer-VALU issue bound: 44 (read-per-lane kernel) / 38 (position kernel) VALU wave-instructions per hashed k-mer, 91-94 % VALU busy (profiles/r01_kernel_stats.md SQ section)",
                           # secondary ceiling: VALU issue = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz lane-ops/s
                           "valu_ceiling": {"instr_per_kmer": ipk, "kmers_per_launch": int(hashed),
                                            "min_ms": round(ipk * hashed / (256 * 4 * 16 * 2.4e9) * 1e3, 4),
                                            "frac": round(ipk * hashed / (256 * 4 * 16 * 2.4e9) * 1e3 / avg_ms, 3)}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not long_mode:
        try:
            from oracle import oracle as O  # noqa: F401  (cpu_baseline leg only)
            G_s = min(db.n_genomes, 16000)
            # bounded DB sample for the CPU probe: the oracle needs the genome-major layout, regenerate decoys of that size
            dk, doff = synth.decoy_sketches(G_s, c=c, device=device, seed=args.seed + 7)
            cb = cpu_baseline(bases, rec_off, n_pairs, read_len, c, k, (dk.cpu().numpy().view(np.uint64), doff.cpu().numpy().astype(np.uint64)))
            t_cpu = n_bases / 1e9 / cb["sketch_gbp_per_s"] + n_total / cb["comparisons_per_s"]
            out["cpu_baseline"] = {"value": round(n_bases / 1e9 / t_cpu, 4), "unit": "Gbp/s", "cores": cb["probe_cores"], "kind": "port",
                                   "sample": cb["sample"], "sketch_gbp_per_s": round(cb["sketch_gbp_per_s"], 4),
                                   "sketch_cores": 1, "genome_comparisons_per_s": round(cb["comparisons_per_s"], 1),
                                   "note": "C++ restatement of the reference CPU path (oracle/); the reference sketches one sample on one thread and probes genomes on all threads"}
        except Exception as e:  # the baseline leg must never sink the GPU measurement
            out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    db.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
