#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X sketch + profile hot path.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its N ranks itself;
                                                            under `python -m torch.distributed.run` it is one of them)

Metric (BASELINE.json): read Gbp/s sketched + genome-comparisons/s profiled, 1 Gbp of 2x150 bp reads vs a GTDB-R220-scale
database (113,104 genome sketches, k=31, c=200).  The unit of work is one SAMPLE through both stages: sketch it (seeding ->
exact dedup/count; reads already resident in HBM) and profile the table against the resident database (containment counts +
coverage vectors back on the host).  One *step* = one pass of the hot path over a batch of `samples_per_step` samples on every
GPU; the batch size is fixed BEFORE the timed region (from the untimed calibration, so that the K timed steps last at least
--min-seconds, default 2 s: a 60 ms region — 20 one-sample steps — let a single scheduler hiccup or an rocm-smi poll move the
result by double-digit percentages) and reported in the line.  `value` = whole-job read Gbp/s through both stages = all bases
of the K steps / the time of the K steps; per-step and per-sample percentiles are printed next to it so that a stall is visible.

Two ways of running the same samples, both measured in every run, `value` taken from the faster one (`mode`; the calibration
decides before the timed region, all ranks agree):
  pipelined   sylph_pipeline_* (csrc/pipeline.hip): C++ sketch worker threads with a context each + one profile thread inside
              the library, a few samples in flight, the way a multi-sample `sylph profile` run overlaps its samples on the rayon
              pool (sketch.rs:313, contain.rs:267-289).  No Python thread takes part in the overlap.
  sequential  one sample at a time on one context: step latency, and every kernel alone on the GPU (`one_step_at_a_time`).

Workloads (--workload): c3 = BASELINE configs[2], the configuration the metric is quoted on (default at N = 1: database on the
one GPU); c4 = configs[3] (default at N > 1: 8 samples per GPU per probe batch); c2 / c5 = configs[1] / [4]; c3r = c3 with ragged
35-151 bp reads and 0.1 % N (reported beside c3, not instead of it).

Multi-GPU: samples are independent units (the reference deals them to its rayon workers).  --db-mode replicate (default
since round 5): every rank holds the whole index (29 GB of 288 GB), sketches and profiles its own samples, no data-path collective at
all; --db-mode shard: every rank holds the postings of one k-mer range — per probe batch the library all-gathers the slice
boundaries, sends every rank its 1/N slice of every table (all-to-all), probes, and sends every hit to the rank that owns its sample
(a second all-to-all); --db-mode genome: north_star's wording — whole genomes per rank (sylph_db_upload_genome_shard), the same
exchange with whole tables as slices (per-rank probe work grows with N: for the A/B).  scaling = weak (per-GPU work fixed).
A single PROCESS over several GPUs (sylph_pipeline_create_multi, what `sylph-hip profile --gpus N` runs) is measured by
tools/multi_gpu_pipeline_bench.py.

After the timed region (untimed): --verify compares the containment results of one more sample, for the sequence-backed genomes
+ a sample of the decoys, against the CPU oracle on the same table; the CPU baseline leg (rank 0, N = 1) times the oracle — the
C++ restatement of the reference's AVX2/rayon path — on a bounded sample of the same inputs.  These two legs are the only places
bench.py touches oracle/.
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
import synth  # noqa: E402  (workload generators: beside bench.py, not in the product package)

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
         "c5": "ONT-like long reads (N50 10 kb, 5 Gbp, 5 % errors, substitution:insertion:deletion = 2:1:1) sketched at c=100 vs GTDB-R220-scale DB at c=200 (BASELINE configs[4])"}


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
    idx_g = torch.arange(glen, dtype=torch.int64, device=device)
    mut_thr = int(0.03 * (1 << 32))
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
                mask = synth._u32(synth.sm64(synth.stream(seed, 2_000_000 + g), idx_g)) < mut_thr   # 97 % identity, substitutions only
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
    verify_set = (v_ids, np.concatenate(v_parts), v_off, (seq_k, seq_off))
    kmers = torch.cat([torch.from_numpy(seq_k.view(np.int64)).to(device), dk])
    goff = torch.cat([torch.from_numpy(seq_off).to(device), doff[1:] + int(seq_off[-1])])
    del dk
    torch.cuda.synchronize()
    t2 = time.time()
    genome_ranges = None
    if db_mode == "shard":      # every rank generated the same database and keeps the postings of its k-mer range
        bounds = S.shard_bounds((2**64 - 1) // c - 1, world)
        db = S.Database(ctx, kmers.data_ptr(), goff.data_ptr(), device_ptrs=True, n_genomes=n_total, shard=(bounds, world, rank))
    elif db_mode == "genome":   # north_star's wording, inside the library (round 5): whole genomes per rank, global ids, the library's exchange
        gb = SH.genome_shard_bounds(goff.cpu().numpy().astype(np.uint64), world)
        genome_ranges = gb
        db = S.Database(ctx, kmers.data_ptr(), goff.data_ptr(), device_ptrs=True, n_genomes=n_total, genome_shard=(gb, world, rank))
    elif db_mode == "genome-py":   # the same cut composed OUTSIDE the library (round 4: unsharded index per rank + torch.distributed all-gathers)
        genome_ranges = SH.genome_shard_ranges(goff.cpu().numpy(), world)
        g0, g1 = int(genome_ranges[rank]), int(genome_ranges[rank + 1])
        k0 = int(goff[g0].item())
        goff_loc = (goff[g0:g1 + 1] - goff[g0]).contiguous()
        db = S.Database(ctx, kmers.data_ptr() + 8 * k0, goff_loc.data_ptr(), device_ptrs=True, n_genomes=g1 - g0)
        ctx.synchronize()
        del goff_loc
    else:
        # tuning experiment (SYLPH_BENCH_DB_CU_MASK=lo:hi): the database — hence the pipeline's profile thread — on a context of its own
        # whose stream is confined to a subset of the CUs, so that the latency-bound probe does not hold wave slots on every CU
        mask = os.environ.get("SYLPH_BENCH_DB_CU_MASK", "")
        dctx = ctx
        if mask:
            dctx = S.Context(device.index)
            dctx.set_option("cu_mask", mask)
            build_database.keep = dctx
        db = S.Database(dctx, kmers.data_ptr(), goff.data_ptr(), device_ptrs=True, n_genomes=n_total)
        dctx.synchronize()
    ctx.synchronize()
    t3 = time.time()
    stats = dict(n_genomes=int(n_total), db_kmers_total=int(kmers.numel()), shard_kmers=int(db.n_kmers), index_gb=round(db.index_bytes / 1e9, 2),
                 seq_backed_generate_and_sketch_s=round(t1 - t0, 2), seq_backed_sketch_s=round(t_sketch, 4),
                 db_build_gbp_per_s=round(n_seq * glen / 1e9 / max(t_sketch, 1e-9), 2), db_build_genomes=int(n_seq),
                 generate_s=round(t2 - t1, 2), db_upload_index_s=round(t3 - t2, 2))
    # NB: no torch.cuda.empty_cache() here — returning tens of GB to the driver (hipFree) queues page-table work
    # that stalls this process's GPU queues for 15-40 ms at random moments over the next few hundred ms.
    del kmers, goff
    stats["genome_ranges"] = [int(x) for x in genome_ranges] if genome_ranges is not None else None
    return db, int(n_total), community, stats, verify_set


def effective_cpus():
    """CPUs this process may really use: os.cpu_count() cut down to the affinity mask and to the cgroup's CPU quota (a container that
    shows 256 hardware threads under `cpu.max` = 16 CPUs runs 256 threads for 6 ms of every 100 ms period and is frozen for the rest)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    q = None
    dirs = ["/sys/fs/cgroup", "/sys/fs/cgroup/cpu"]
    try:
        for ln in open("/proc/self/cgroup"):
            path = ln.strip().split(":")[-1]
            if path.startswith("/") and path != "/":
                dirs += ["/sys/fs/cgroup" + path, "/sys/fs/cgroup/cpu" + path]
    except Exception:
        pass
    for d in dirs:
        try:
            a, per = open(d + "/cpu.max").read().split()[:2]
            if a != "max":
                q = min(q or 1e9, float(a) / float(per))
        except Exception:
            pass
        try:
            quota, per = int(open(d + "/cpu.cfs_quota_us").read()), int(open(d + "/cpu.cfs_period_us").read())
            if quota > 0 and per > 0:
                q = min(q or 1e9, quota / per)
        except Exception:
            pass
    if q:
        n = min(n, max(1, int(np.ceil(q))))
    return n


def cpu_baseline(bases, rec_off, n_records, paired, c, k, db_host, whole):
    """Oracle (C++ restatement of the reference CPU path) on this box's host cores.  whole: the WHOLE sample (every read on one
    thread — the reference sketches one sample per thread, sketch.rs:313,371 — and every genome of the database on all threads,
    contain.rs:284): nothing extrapolated, and the results double as the full-size parity check (verify).  Otherwise a bounded
    sample, extrapolated (boxes without the host memory for the genome-major database).  Reads of any shape: `bases` / `rec_off` are
    the sample's device arrays (pairs interleaved mate 1, mate 2; long reads single-end)."""
    from oracle import oracle as O
    cores = effective_cpus()                       # (threads beyond the container's CPU quota only get the whole process throttled)
    n_s = n_records
    if not whole:                                  # about 600 Mbp from the front of the sample (whole pairs)
        n_s = int(torch.searchsorted(rec_off, torch.tensor([600_000_000], dtype=rec_off.dtype, device=rec_off.device)).item())
        n_s = max(2, min(n_records, n_s)) & ~1
    ho = rec_off[: n_s + 1].cpu().numpy().astype(np.uint64)
    nb = int(ho[-1])
    hb = bases[:nb].cpu().numpy()
    mode = O.MODE_AVX2_FAST if O.lib().orc_has_avx2() else O.MODE_SCALAR
    t = time.perf_counter()
    sk = O.sketch_reads(hb, ho, c=c, k=k, mode=mode, paired=paired)
    t_sketch = time.perf_counter() - t
    sketch_gbps = nb / t_sketch / 1e9                           # one sample = one thread in the reference (sketch.rs:313,371)
    dbk, dbo = db_host
    ls = O.LoadedSample(sk["kmers"], sk["counts"])
    cc, cov, t_probe = ls.probe(dbk, dbo, n_threads=cores)      # genomes in parallel on all cores (contain.rs:284)
    ls.close()
    G = len(dbo) - 1
    shape = f"{n_s // 2} read pairs" if paired else f"{n_s} reads"
    what = (f"the WHOLE sample: {shape} ({nb / 1e6:.0f} Mbp) sketched on 1 thread in {t_sketch:.2f} s "
            f"({'AVX2 intrinsics' if mode == O.MODE_AVX2_FAST else 'scalar'}), its {len(sk['kmers'])}-entry table probed against all {G} genomes "
            f"({len(dbk) / 1e6:.0f} M k-mers) on {cores} threads in {t_probe:.2f} s; nothing extrapolated") if whole else (
            f"sketch: first {shape} ({nb / 1e6:.0f} Mbp of the sample) on 1 thread "
            f"({'AVX2 intrinsics' if mode == O.MODE_AVX2_FAST else 'scalar'}); probe: {G} of the genomes ({len(dbk) / 1e6:.0f} M k-mers) on {cores} "
            f"threads vs the {len(sk['kmers'])}-entry sample table; both rates EXTRAPOLATED linearly to the full workload")
    return dict(sketch_gbp_per_s=sketch_gbps, comparisons_per_s=G / t_probe, sketch_cores=1, probe_cores=cores, sample=what,
                sketch_s=t_sketch, probe_s=t_probe, table=sk, contain_count=cc, cov=cov, genome_off=dbo)


def end_to_end_from_files(bases, n_pairs_total, read_len, n_pairs):
    """Untimed w.r.t. `value`: the first n_pairs pairs of a read set written as FASTQ files on local disk (plain, and `gzip -1`:
    one ordinary single-member .gz per mate), then the product's own command on them — `sylph-hip sketch -1 .. -2 ..` — wall clock
    of the WHOLE command (process start, GPU bring-up, parsing / inflating, H2D, kernels, .sylsp written), and the per-sample
    times the command logs (the first sample of a process pays the bring-up; the later ones of a 4-sample command are warm)."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import feed_bench as FB
    n_pairs = int(min(n_pairs, n_pairs_total))
    d = tempfile.mkdtemp(prefix="sylph_e2e_")
    try:
        hb = bases[: n_pairs * 2 * read_len].cpu().numpy().reshape(n_pairs, 2, read_len)
        FB.write_fastq(f"{d}/s_1.fq", np.ascontiguousarray(hb[:, 0, :]).reshape(-1), read_len)
        FB.write_fastq(f"{d}/s_2.fq", np.ascontiguousarray(hb[:, 1, :]).reshape(-1), read_len)
        del hb
        gz = [subprocess.Popen(["gzip", "-1", "-k", "-f", f"{d}/s_{m}.fq"]) for m in (1, 2)]
        for i in range(1, 4):
            for m in (1, 2):
                os.symlink(f"{d}/s_{m}.fq", f"{d}/p{i}_{m}.fq")
        os.symlink(f"{d}/s_1.fq", f"{d}/p0_1.fq")
        os.symlink(f"{d}/s_2.fq", f"{d}/p0_2.fq")
        gbp = 2 * n_pairs * read_len / 1e9
        env = dict(os.environ)
        env.pop("SYLPH_HIP_EXACT_DEDUP", None)          # default flags, as a user runs them: pairs behind the cuckoo filter (--fpr 1e-4)
        exe = os.path.join(ROOT, "sylph_amd", "sylph-hip")

        def run(args, settle=2.0):
            # every timed command starts on a quiet device: `settle` seconds after the previous process has exited (the driver wipes a
            # process's HBM behind its exit — 7-13 GB here — and the next process's first allocations wait for that: what it costs a command
            # started back to back is reported as `back_to_back`)
            time.sleep(settle)
            t = time.perf_counter()
            p = subprocess.run([exe, "sketch", *args, "-d", f"{d}/out"], capture_output=True, text=True, env=env, timeout=600)
            dt = time.perf_counter() - t
            if p.returncode != 0:
                raise RuntimeError(p.stderr[-500:])
            per = [float(ln.split(" in ")[1].split(" s")[0]) for ln in p.stderr.split("\n") if "timing:" in ln]
            return dt, per
        out = {"pairs_per_sample": n_pairs, "gbp_per_sample": round(gbp, 4), "host_threads": os.cpu_count(), "host_cpus_usable": effective_cpus(),
               "what": "`sylph-hip sketch` with default flags (pairs deduplicated behind the cuckoo filter, --fpr 1e-4) on FASTQ files in a temporary "
                       "directory: whole-command wall clock and the per-sample times it logs"}
        # untimed: the first process on a box loads the binary, the libraries and the GPU runtime's files from disk
        FB.write_fastq(f"{d}/w_1.fq", np.ascontiguousarray(bases[: 2000 * read_len].cpu().numpy()), read_len)
        FB.write_fastq(f"{d}/w_2.fq", np.ascontiguousarray(bases[2000 * read_len: 4000 * read_len].cpu().numpy()), read_len)
        run(["-1", f"{d}/w_1.fq", "-2", f"{d}/w_2.fq"], settle=0.0)
        out["settle_seconds_before_each_command"] = 2.0
        dt, per = run(["-1", *[f"{d}/p{i}_1.fq" for i in range(4)], "-2", *[f"{d}/p{i}_2.fq" for i in range(4)], "-t", "1"])
        out["plain_four_samples_one_command"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(4 * gbp / dt, 3),
                                                 "sample_gbp_per_s_in_order": [round(gbp / x, 2) for x in per]}
        def three(args):      # the GPU runtime's start-up varies by 0.1-0.2 s from one process to the next: the median of three runs, all three kept
            runs = [run(args) for _ in range(3)]
            runs.sort(key=lambda r: r[0])
            return runs[1][0], runs[1][1], [round(r[0], 3) for r in runs]
        dt, per, all3 = three(["-1", f"{d}/s_1.fq", "-2", f"{d}/s_2.fq"])
        out["plain_one_sample"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(gbp / dt, 3),
                                   "sample_gbp_per_s": round(gbp / per[0], 2) if per else None, "command_seconds_three_runs": all3}
        for g in gz:
            g.wait()
        out["gz_bytes_per_file"] = os.path.getsize(f"{d}/s_1.fq.gz")
        dt, per, all3 = three(["-1", f"{d}/s_1.fq.gz", "-2", f"{d}/s_2.fq.gz"])
        out["gz_one_sample"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(gbp / dt, 3),
                                "sample_gbp_per_s": round(gbp / per[0], 2) if per else None, "command_seconds_three_runs": all3,
                                "what": "one ordinary (single-member) gzip -1 file per mate: the COMPRESSED bytes go to the device, which inflates them "
                                        "(csrc/inflate.hip; round 6); median of three runs"}
        dt, per = run(["-1", f"{d}/s_1.fq.gz", "-2", f"{d}/s_2.fq.gz"], settle=0.0)
        out["gz_one_sample"]["back_to_back"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(gbp / dt, 3),
                                                "what": "the same command started the moment the previous one exited (no settle time)"}
        env["SYLPH_HIP_INFLATE_DEVICE"] = "0"
        dt, per = run(["-1", f"{d}/s_1.fq.gz", "-2", f"{d}/s_2.fq.gz"])
        env.pop("SYLPH_HIP_INFLATE_DEVICE")
        out["gz_one_sample"]["inflated_on_the_host"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(gbp / dt, 3),
                                                        "what": "the same command with SYLPH_HIP_INFLATE_DEVICE=0: host/pgunzip.cpp on the host's threads (round 5's road)"}
        # the whole 1 Gbp sample, through the device's inflate and as plain text: the two .sylsp files hold the same table
        # (types.rs:145-155: u64 length, then (u64 k-mer, u32 count) entries lead the file)
        def table(path):
            raw = open(path, "rb").read()
            n = int.from_bytes(raw[:8], "little")
            t = np.frombuffer(raw, dtype=np.dtype([("k", "<u8"), ("c", "<u4")]), count=n, offset=8)
            o = np.argsort(t["k"], kind="stable")
            return t["k"][o], t["c"][o]
        try:
            (ka, ca), (kb, cb) = table(f"{d}/out/s_1.fq.paired.sylsp"), table(f"{d}/out/s_1.fq.gz.paired.sylsp")
            out["gz_one_sample"]["verify"] = {"table_entries": int(len(ka)), "gz_table_equals_plain_table": bool(len(ka) == len(kb) and (ka == kb).all() and (ca == cb).all())}
        except Exception as e:
            out["gz_one_sample"]["verify"] = {"error": str(e)[:200]}
        for i in range(4):
            for m in (1, 2):
                os.symlink(f"{d}/s_{m}.fq.gz", f"{d}/g{i}_{m}.fq.gz")
        dt, per = run(["-1", *[f"{d}/g{i}_1.fq.gz" for i in range(4)], "-2", *[f"{d}/g{i}_2.fq.gz" for i in range(4)], "-t", "1"])
        out["gz_four_samples_one_command"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(4 * gbp / dt, 3),
                                              "sample_gbp_per_s_in_order": [round(gbp / x, 2) for x in per]}
        # the same four samples over two sample threads (`-t 2`: a second engine — context, stream, warm-up — comes up beside the first; the
        # reference's default is 3 threads, sketch.rs:313)
        dt, per = run(["-1", *[f"{d}/g{i}_1.fq.gz" for i in range(4)], "-2", *[f"{d}/g{i}_2.fq.gz" for i in range(4)], "-t", "2"])
        out["gz_four_samples_one_command"]["two_sample_threads"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(4 * gbp / dt, 3)}
        dt, per = run(["-1", *[f"{d}/p{i}_1.fq" for i in range(4)], "-2", *[f"{d}/p{i}_2.fq" for i in range(4)], "-t", "2"])
        out["plain_four_samples_one_command"]["two_sample_threads"] = {"command_seconds": round(dt, 3), "command_gbp_per_s": round(4 * gbp / dt, 3)}
        # The same files through the CPU path as the reference runs it (sketch.rs:313, :371: one rayon worker per sample; needletail +
        # flate2 on that thread): oracle/'s fast seeding + its model of the default pair dedup behind a zlib reader, ONE thread per
        # sample, as many samples side by side as the box may use CPUs.  kind = "port": the oracle, timed, never the product.
        try:
            from oracle import oracle as O
            cpus = effective_cpus()
            cpu = {"kind": "port", "cores_usable": cpus, "threads_per_sample": 1,
                   "what": "oracle/ (C++ restatement of the reference's CPU path: AVX2 seeding, filter-model dedup, zlib reader) on the SAME files, one thread per "
                           "sample, min(samples, usable CPUs) samples at a time — the reference's own parallelism (sketch.rs:313, :371)"}
            for kind, ext in (("plain", "fq"), ("gz", "fq.gz")):
                for n in (1, 4, 16):
                    try:
                        avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
                    except Exception:
                        avail = 0
                    # (a sample is ~2.3 B per base in host memory while its two mates are interleaved: never risk the box for a baseline)
                    if n > 1 and (n > 2 * cpus or min(n, cpus) * gbp * 2.6e9 > avail / 2):
                        cpu[f"{kind}_{n}_samples"] = {"skipped": "not enough CPUs or host memory for that many samples side by side"}
                        continue
                    f1 = [f"{d}/s_1.{ext}"] * n
                    f2 = [f"{d}/s_2.{ext}"] * n
                    r = O.sketch_files(f1, f2, c=200, k=31, fpr=1e-4, threads=min(n, cpus))
                    cpu[f"{kind}_{n}_samples"] = {"wall_seconds": round(r["wall_seconds"], 3), "gbp_per_s": round(n * gbp / r["wall_seconds"], 3),
                                                  "threads": min(n, cpus), "seconds_per_sample_mean": round(float(np.mean(r["seconds"])), 3)}
                    if n == 16 and r["wall_seconds"] > 40:
                        break
            out["cpu_baseline_from_files"] = cpu
        except Exception as e:
            out["cpu_baseline_from_files"] = {"error": str(e)[:300]}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def verify_against_oracle(ctx, res, G, sample_ptrs, verify_set, device):
    """Untimed: the last step's first sample — contain_count and the sorted coverage vector of every genome of the verify set
    (all sequence-backed genomes + 2,000 decoys) against the CPU oracle probing the same table."""
    from oracle import oracle as O
    dk, dc, n = sample_ptrs
    sk = SH.device_view(dk, n, torch.int64, device).cpu().numpy().view(np.uint64)
    sc = SH.device_view(dc, n, torch.int32, device).cpu().numpy().view(np.uint32)
    v_ids, v_k, v_off = verify_set[:3]
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


def csrc_fingerprint():
    """sha256 over the kernel sources: the PMC traffic figures in profiles/ are only quoted for the code they were measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sylph_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("SYLPH_BENCH_WORKLOAD", "auto"), choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--db-mode", default=os.environ.get("SYLPH_BENCH_DB_MODE", "auto"), choices=["auto", "replicate", "shard", "genome", "genome-py"])
    ap.add_argument("--shard-reduce", default=os.environ.get("SYLPH_BENCH_SHARD_REDUCE", "alltoall"), choices=["alltoall", "allgather"],
                    help="sharded databases (--db-mode shard | genome): how the hits reach the rank that owns the sample — all-to-all (default), or ONE all-gather "
                         "of padded blocks (north_star's wording; csrc/shard_plan.h plan_hits_gather)")
    ap.add_argument("--mode", default=os.environ.get("SYLPH_BENCH_MODE", "auto"), choices=["auto", "pipelined", "sequential"],
                    help="which way of running the samples `value` is taken from (auto: the faster one of the untimed calibration)")
    ap.add_argument("--min-seconds", type=float, default=float(os.environ.get("SYLPH_BENCH_MIN_SECONDS", "2.0")),
                    help="the K timed steps last at least this long: samples_per_step is sized for it before the timed region")
    ap.add_argument("--samples-per-step", type=int, default=0, help="samples per GPU per step (default: sized from --min-seconds)")
    ap.add_argument("--probe-batch", type=int, default=0, help="sample tables per probe launch (default: the workload's; c4: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-bounded", action="store_true", help="time the CPU restatement on a bounded sample (extrapolated) instead of the whole sample")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-fed (PCIe-inclusive) leg")
    ap.add_argument("--no-kernel-timers", action="store_true", help="skip the in-library HIP-event kernel timers (no roofline objects)")
    ap.add_argument("--all-kernel-timers", action="store_true", help="time every kernel family inside the timed region too (rounds 1-4 did; costs ~1 %% pipelined, ~4 %% one at a time)")
    ap.add_argument("--sketch-workers", type=int, default=int(os.environ.get("SYLPH_BENCH_SKETCH_WORKERS", "0")), help="sketch worker threads of the pipeline, each with its own context/stream (default 3)")
    ap.add_argument("--pipeline-depth", type=int, default=int(os.environ.get("SYLPH_BENCH_PIPELINE_DEPTH", "0")), help="samples in flight in the pipeline (default: workers + 5; sharded: two probe batches)")
    ap.add_argument("--no-files-leg", action="store_true", help="skip the leg that runs `sylph-hip sketch` on FASTQ files (plain, gzip)")
    ap.add_argument("--files-leg-pairs", type=int, default=3_333_334, help="read pairs per sample of that leg (default: the workload's own sample, 1 Gbp)")
    ap.add_argument("--no-packed-leg", action="store_true", help="skip the leg with the reads resident as packed 2-bit")
    ap.add_argument("--no-filter-leg", action="store_true", help="skip the leg with sylph's default pair dedup (cuckoo filter, --fpr 1e-4)")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the leg of the mode `value` is NOT taken from")
    ap.add_argument("--main-dedup-fpr", type=float, default=0.0, help="profiling runs only: EVERY leg's sessions get sylph's default pair dedup (the cuckoo "
                                                                      "filter of csrc/a10.hip) with this --fpr; switches the exact-set verify and the filter leg off")
    ap.add_argument("--seed", type=int, default=20250711)
    ap.add_argument("--sweep", default="", help="tuning runs only: JSON list of pipeline configurations ({name, workers, depth, max_batch, options{}, "
                                                "pipe_options{}}) measured one after the other on the workload of this run, after the timed region; "
                                                "results go into the line's `sweep` object (never into `value`)")
    args = ap.parse_args()
    if args.main_dedup_fpr:
        args.no_verify = args.no_filter_leg = True

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

    def agree(value, op="max"):            # one number every rank ends up with
        if dist is None:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.MIN)
        return float(t.item())

    c, k, read_len = 200, 31, 150
    n_pairs, _, _, _, _, spb, n_sets = WORKLOADS[wl]
    if args.probe_batch:
        spb = args.probe_batch
    n_sets = max(n_sets, spb)
    long_mode = wl == "c5"
    c_reads = 100 if long_mode else c        # reads may be sketched denser than the DB (contain.rs:562-568,616-623)
    # The main context launches on a torch-owned stream, so torch-side generation, the library's kernels on this context and the
    # timing events are stream-ordered without cross-queue synchronisation.
    # (SYLPH_BENCH_STREAM_PRIORITY=high|low: the main — in the pipeline: the profile thread's — stream at another priority; A/B for VERDICT r05 #5)
    prio = {"high": -1, "low": 0, "": 0}[os.environ.get("SYLPH_BENCH_STREAM_PRIORITY", "")]
    tstream = torch.cuda.Stream(device=device, priority=prio)
    torch.cuda.set_stream(tstream)
    ctx = S.Context(local, stream=tstream.cuda_stream)
    ctx.set_option("shard_reduce", args.shard_reduce)
    ctx_options = [kv.split("=", 1) for kv in filter(None, os.environ.get("SYLPH_BENCH_CTX_OPTIONS", "").split(","))]   # tuning experiments only
    for kv in ctx_options:
        ctx.set_option(*kv)

    # auto = replicas (round 5): the samples are the independent units (the reference deals them to its rayon workers: contain.rs:252-295,
    # sketch.rs:313,371), the 29 GB index fits every GPU, and no data-path collective is needed — `--db-mode shard` (k-mer ranges) and
    # `--db-mode genome` (north_star's wording) run the library's exchange instead, for databases that do not fit and for the A/B
    db_mode = args.db_mode if args.db_mode != "auto" else "replicate"
    log(f"[bench] building workload {wl} on {world} GPU(s), db {db_mode} ...")
    db, n_total, community, dbstats, verify_set = build_database(ctx, device, wl, c, k, args.seed, rank, world, db_mode)
    comm, comm_kind, fallbacks = None, None, []

    def agreed_failure(failed):            # a rank that failed takes every rank down the same fallback
        return bool(agree(1 if failed else 0))

    if db_mode in ("shard", "genome"):
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
    read_sets = []                           # distinct samples, rotated so that the probe is never cache-warm
    for i in range(n_sets):
        sd = args.seed + 1_000_003 * (rank + 1) + 7919 * i
        if long_mode:
            bases, rec_off = synth.long_reads(community, 5_000_000_000, seed=sd)
            n_records = rec_off.numel() - 1
            cuts = [0]                       # a push holds < 2^32 bases: batches of whole reads
            while cuts[-1] < n_records:
                nxt = int(torch.searchsorted(rec_off, rec_off[cuts[-1]] + 3_000_000_000).item())
                cuts.append(max(cuts[-1] + 1, min(nxt, n_records)))
            batches, keep = [], []
            for a, b in zip(cuts[:-1], cuts[1:]):
                o = (rec_off[a:b + 1] - rec_off[a]).contiguous()
                keep.append(o)
                batches.append((bases.data_ptr() + int(rec_off[a].item()), o.data_ptr(), b - a, int(o[-1].item())))
            read_sets.append(dict(bases=bases, rec_off=rec_off, keep=keep, n_bases=int(rec_off[-1].item()), n_records=n_records, batches=batches))
        else:
            if wl == "c3r":
                bases, rec_off = synth.ragged_paired_reads(community, n_pairs, seed=sd)
            else:
                bases, rec_off = synth.paired_reads(community, n_pairs, read_len=read_len, seed=sd)
            nb = int(rec_off[-1].item())
            read_sets.append(dict(bases=bases, rec_off=rec_off, n_bases=nb, n_records=2 * n_pairs,
                                  batches=[(bases.data_ptr(), rec_off.data_ptr(), 2 * n_pairs, nb)]))
    torch.cuda.synchronize()
    del community
    n_bases = float(np.mean([r["n_bases"] for r in read_sets]))
    log(f"[bench] db {dbstats}; {n_sets} read sets of {n_bases / 1e9:.3f} Gbp generated in {time.time() - t0:.1f}s")

    # ---- the two ways of running samples ------------------------------------------------------------------------------------
    n_workers = max(1, args.sketch_workers or 3)
    depth = max(1, args.pipeline_depth or (2 * spb if comm is not None else max(n_workers + 5, spb + n_workers)))
    if comm is not None:
        depth = max(depth, spb)
    sample_no = [0]

    from sylph_amd.binding import ENC_2BIT as _ENC_2BIT, ENC_ASCII as _ENC_ASCII, MEM_DEVICE as _MEM_DEVICE
    active_sets = [read_sets]                # (the packed-input leg swaps in the 2-bit copies of the same read sets)

    def next_read_set():
        rs = active_sets[0][sample_no[0] % n_sets]
        sample_no[0] += 1
        return rs

    pipe_options = [kv.split("=", 1) for kv in filter(None, os.environ.get("SYLPH_BENCH_PIPE_OPTIONS", "").split(","))]   # tuning experiments only

    def make_pipeline():
        p = S.Pipeline(db, c=c_reads, k=k, paired=not long_mode, n_workers=n_workers, depth=depth, max_batch=spb if comm is not None else max(spb, 8),
                       comm=comm)
        for kv in ctx_options + pipe_options:
            p.set_option(*kv)
        if args.main_dedup_fpr:
            p.set_option("dedup_fpr", repr(args.main_dedup_fpr))
        return p

    pipe_box = [None]

    def run_pipelined(n, stamps=None, rows=None):
        """n samples through sylph_pipeline_*; stamps: completion time of every sample (perf_counter)"""
        p = pipe_box[0]
        sub = done = 0
        while done < n:
            while sub < n and p.outstanding < depth:
                rs = next_read_set()
                if not p.submit_device(rs["batches"], tag=sub, enc=rs.get("enc", _ENC_ASCII)):
                    raise RuntimeError("pipeline refused a sample below its depth")
                sub += 1
                if sub == n and comm is not None:
                    p.flush()                # the last probe batch of this call may be a partial one (the same on every rank)
            r = p.next(views=False)
            done += 1
            if stamps is not None:
                stamps.append(time.perf_counter())
            if rows is not None:
                t = r["t"]
                rows.append((t[2] - t[1], t[4] - t[3], r["n_table"], r["dup_removed"], r["n_covs"], r["probe_batch"]))

    filter_box = [bool(args.main_dedup_fpr)]    # the filter leg: sessions get the reference's default pair dedup (dedup_fpr 1e-4)

    def sketch_inline(rs):
        sk = S.ReadSketcher(ctx, c=c_reads, k=k, paired=not long_mode, dedup_fpr=1e-4 if filter_box[0] else 0.0)
        if len(rs["batches"]) == 1:
            sk.set_option("borrow_until_finish", 1)     # the resident read sets outlive every session
        for bptr, optr, nrec, nb in rs["batches"]:
            if rs.get("enc", _ENC_ASCII) == _ENC_ASCII:
                sk.push_device(bptr, optr, nrec, nb)
            else:
                sk.push_enc(bptr, optr, nb, _MEM_DEVICE, rs["enc"], n_records=nrec)
        return sk, sk.finish_device()

    def run_sequential(n, stamps=None, rows=None, keep_last=None):
        """n samples one at a time on the main context (probe batches of spb: their sketches run one after the other first)"""
        done = 0
        while done < n:
            m = min(spb, n - done)
            t_a = time.perf_counter()
            sess = [sketch_inline(next_read_set()) for _ in range(m)]
            t_b = time.perf_counter()
            refs = [(dk, dc, nt) for _, (dk, dc, nt, _) in sess]
            if comm is not None:
                res = db.contain_batch_sharded(comm, refs, device_ptrs=True)
            elif db_mode == "genome-py":
                res = SH.contain_batch_genome_sharded(dist, db, np.array(dbstats["genome_ranges"]), refs, device)
            else:
                res = db.contain_batch(refs, device_ptrs=True)      # borrowed pinned views
            t_c = time.perf_counter()
            if keep_last is not None:
                keep_last["res"] = tuple(np.array(x) for x in res)
                keep_last["table"] = sess[0][1][:3]
                keep_last["occ"] = [int(SH.device_view(dc, nt, torch.int32, device).sum().item()) + dup for _, (dk, dc, nt, dup) in sess]
                keep_last["sessions"] = [s for s, _ in sess]
            else:
                for s, _ in sess:
                    s.close()
            for j, (_, (dk, dc, nt, dup)) in enumerate(sess):
                if stamps is not None:
                    stamps.append(t_c)
                if rows is not None:
                    rows.append(((t_b - t_a) / m, (t_c - t_b) / m, nt, dup, len(res[2]) // m, m))
            done += m

    runners = {"pipelined": run_pipelined, "sequential": run_sequential}

    # ---- untimed: first-use allocations of every context's pool, lazy kernel loading, then the calibration -------------------
    pipe_box[0] = make_pipeline()
    if comm is not None and world > 1:
        # the first sharded batches ever run on this node: if the exchange fails on any rank, every rank drops to the replicated
        # database (no data-path collective) and says so in the result line
        err = None
        try:
            run_sequential(spb)
        except Exception as e:
            err = e
        if agreed_failure(err is not None):
            fallbacks.append(f"sharded exchange ({comm_kind}): {err}")
            log(f"[bench] rank {rank}: sharded step failed ({err}); falling back to the replicated database")
            try:
                pipe_box[0].close()
                comm.close()
                db.close()
            except Exception:
                pass
            comm, comm_kind, db_mode = None, None, "replicate"
            db, n_total, _, dbstats, verify_set = build_database(ctx, device, wl, c, k, args.seed, rank, world, db_mode)
            pipe_box[0] = make_pipeline()
    run_sequential(2 * spb)
    if db_mode != "genome-py":
        run_pipelined(max(2 * depth, 2 * spb))
    torch.cuda.synchronize()

    def measure(mode, n):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        runners[mode](n)
        torch.cuda.synchronize()
        return agree((time.perf_counter() - t_s) / n)      # slowest rank's seconds per sample

    # ~0.25 s per mode, sized from one probe batch timed first (all ranks agree on the count; a debug run with several ranks on
    # one GPU and host-copy collectives is orders of magnitude slower per batch than the real thing)
    t_batch = measure("sequential", spb) * spb
    n_cal = spb * int(min(400, max(2, round(0.25 / max(t_batch, 1e-6)))))
    cal = {"samples_per_mode": n_cal}
    for mode in ("pipelined", "sequential"):
        if db_mode == "genome-py" and mode == "pipelined":
            continue
        if args.mode in ("auto", mode) or not args.no_second_leg:
            cal[mode + "_ms_per_sample"] = round(measure(mode, n_cal) * 1e3, 4)
    if db_mode == "genome-py":
        mode = "sequential"          # the genome-sharded arm is a step-at-a-time composition (sylph_amd/shard.py), not a pipeline
    elif args.mode != "auto":
        mode = args.mode
    else:
        mode = "pipelined" if cal["pipelined_ms_per_sample"] <= cal["sequential_ms_per_sample"] else "sequential"
    other = "sequential" if mode == "pipelined" else "pipelined"
    est = cal[mode + "_ms_per_sample"] * 1e-3
    # samples per step: a multiple of the probe batch, sized so that the K timed steps last --min-seconds
    if args.samples_per_step:
        sps = max(spb, args.samples_per_step // spb * spb)
    else:
        sps = int(agree(max(1, int(np.ceil(args.min_seconds * 1.1 / max(1, args.steps) / est / spb))) * spb))
    log(f"[bench] calibration {cal}: value from the {mode} mode, {sps} sample(s) per step")

    def timed(mode, n_steps, per_step, only=None):
        """-> elapsed seconds (max over ranks), per-step seconds, per-sample completion stamps, rows, kernel families.
        `only`: the kernel families the library times with HIP events in this region (None: all of them).  Every timed family costs two
        event records per launch group on its stream — all of them together 1 % of a pipelined sample and 4 % of a sample run alone
        (profiles/r05_ab_timers.txt) — so the regions a rate is quoted on time the dominant kernel only; the other families' durations come
        from a pass of their own ("kernel_ms_pass")."""
        profiled = [pipe_box[0]] if mode == "pipelined" else [ctx]
        for o in profiled:
            o.set_option("profile_only", "all" if (only is None or args.all_kernel_timers) else only)
            o.profile(not args.no_kernel_timers)
        if comm is not None:
            db.exchange_stats(reset=True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        stamps, rows, bounds = [], [], [t_start]
        for _ in range(n_steps):
            # (pipelined: the pipeline drains at the end of every step — its last samples have nothing to overlap with; with
            #  hundreds of samples per step that tail is below one per cent, and a step stays a closed unit of work)
            runners[mode](per_step, stamps, rows)
            bounds.append(time.perf_counter())
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = agree(time.perf_counter() - t_start)
        fam = {}
        for f in ("seeds", "compact", "annotate", "sort", "replay", "a10", "probe", "assemble", "exchange"):   # HIP events on the launch streams, all contexts
            fam[f] = profiled[0].kernel_stats(f) if not args.no_kernel_timers else (0.0, 0)
        if comm is not None:
            fam["_exchange_totals"] = db.exchange_stats()
        for o in profiled:
            o.profile(False)
            o.set_option("profile_only", "all")
        return elapsed, list(np.diff(bounds)), list(np.diff([t_start] + stamps)), rows, fam

    for _ in range(args.warmup):
        runners[mode](sps)
    elapsed, step_s, gaps, rows, fam = timed(mode, args.steps, sps, only="seeds,exchange" if comm is not None else "seeds")
    # the other kernel families in the same mix: a shorter pass of the same mode with every family timed (never the quoted rate)
    fam_pass = None
    if not args.no_kernel_timers and not args.all_kernel_timers and not args.no_second_leg:
        st_k = max(1, min(args.steps, 3))
        per_k = max(spb, int(0.4 / max(est, 1e-6) / st_k / spb) * spb)
        e_k, _, _, r_k, f_k = timed(mode, st_k, per_k)
        fam_pass = {"fam": f_k, "rows": r_k, "ms_per_sample": round(e_k / (st_k * per_k) * 1e3, 4), "samples": st_k * per_k}
    second = None
    if not args.no_second_leg and db_mode != "genome-py":               # the same samples the other way, ~0.6 s of them
        n2 = max(spb, int(0.6 / max(cal.get(other + "_ms_per_sample", 1.5) * 1e-3, 1e-6) / spb) * spb)
        steps2 = max(1, min(args.steps, 4))
        per2 = max(spb, n2 // steps2 // spb * spb)
        second = (other, per2) + timed(other, steps2, per2, only="seeds,probe")

    last = {}
    run_sequential(spb, keep_last=last)      # untimed: seed occurrences for the roofline's algorithmic bytes, results for --verify

    def leg_summary(mode_name, per_step, n_steps, elapsed_s, step_list, gap_list, rws, fm):
        n_samples = per_step * n_steps
        s_per_sample = elapsed_s / n_samples
        workers_busy = n_workers if mode_name == "pipelined" else 1
        d = {"mode": mode_name, "samples_per_step": per_step, "steps": n_steps, "timed_region_s": round(elapsed_s, 4),
             "ms_per_step": round(elapsed_s / n_steps * 1e3, 3), "ms_per_sample": round(s_per_sample * 1e3, 4),
             "value": round(world * n_bases / 1e9 / s_per_sample, 3),
             "step_ms": {"p50": round(pct(step_list, 0.5) * 1e3, 3), "p90": round(pct(step_list, 0.9) * 1e3, 3), "max": round(max(step_list) * 1e3, 3)},
             # time between consecutive sample completions on rank 0 (a probe batch completes together: zeros inside a batch)
             "sample_interval_ms": {"p50": round(pct(gap_list, 0.5) * 1e3, 4), "p90": round(pct(gap_list, 0.9) * 1e3, 4),
                                    "p99": round(pct(gap_list, 0.99) * 1e3, 4), "max": round(max(gap_list) * 1e3, 4)},
             # stage times as the sample saw them (wall clock inside its stage; pipelined: stages of different samples overlap)
             "sketch_ms": round(float(np.mean([r[0] for r in rws])) * 1e3, 3), "profile_ms": round(float(np.mean([r[1] for r in rws])) * 1e3, 3),
             "probe_batch_mean": round(float(np.mean([r[5] for r in rws])), 2),
             "kernel_ms": {f: (round(v[0] / max(1, v[1]), 4), int(v[1])) for f, v in fm.items() if not f.startswith("_") and v[1]}}
        if fm.get("_exchange_totals", (0,))[0]:
            nb, tb, hb = fm["_exchange_totals"]
            xt = fm.get("exchange", (0.0, 0))
            # the two payload all-to-alls of a probe batch on THIS rank (rank 0): bytes sent to the other ranks, HIP-event time
            d["exchange"] = {"probe_batches": nb, "table_slice_bytes_sent_per_batch": int(tb / nb), "hit_bytes_sent_per_batch": int(hb / nb),
                             "all_to_all_ms_per_batch": round(xt[0] / nb, 4) if xt[1] else None}
        d["sketch_gbp_per_s"] = round(world * n_bases / 1e9 / max(d["sketch_ms"] * 1e-3 / workers_busy, 1e-9), 3)
        d["genome_comparisons_per_s"] = round(world * n_total / max(d["profile_ms"] * 1e-3, 1e-9), 1)
        return d

    main_leg = leg_summary(mode, sps, args.steps, elapsed, step_s, gaps, rows, fam)
    if fam_pass is not None:
        main_leg["kernel_ms_pass"] = {"what": "every kernel family timed (HIP event pairs), a separate shorter pass of the same mode: the rate it cost",
                                      "samples": fam_pass["samples"], "ms_per_sample": fam_pass["ms_per_sample"],
                                      "kernel_ms": {f: (round(v[0] / max(1, v[1]), 4), int(v[1])) for f, v in fam_pass["fam"].items() if not f.startswith("_") and v[1]}}
    # ---- the same samples with the reads resident as the packed 2-bit stream (SYLPH_ENC_2BIT: what the CLI's feed pushes; a
    # quarter of the bytes, no ASCII -> 2-bit conversion in the seeding kernel).  Reported beside `value`, never instead of it:
    # SURVEY 8d's 1.085 B/base is the ASCII input.
    packed_leg = None
    if not args.no_packed_leg and not long_mode and wl != "c3r" and comm is None and db_mode != "genome-py":
        try:
            packed_sets = []
            for rs in read_sets:
                b = rs["bases"][:rs["n_bases"]]
                pad = (-b.numel()) % 4
                if pad:
                    b = torch.cat([b, torch.full((pad,), 65, dtype=torch.uint8, device=device)])
                cds = (((b >> 1) ^ (b >> 2)) & 3).view(-1, 4)          # ACGT only in these read sets (BYTE_TO_SEQ codes)
                pk = ((cds[:, 0] << 6) | (cds[:, 1] << 4) | (cds[:, 2] << 2) | cds[:, 3]).contiguous()
                packed_sets.append(dict(bases=pk, rec_off=rs["rec_off"], n_bases=rs["n_bases"], n_records=rs["n_records"], enc=_ENC_2BIT,
                                        batches=[(pk.data_ptr(), rs["rec_off"].data_ptr(), rs["n_records"], rs["n_bases"])]))
                del cds
            torch.cuda.synchronize()
            active_sets[0] = packed_sets
            runners["pipelined"](2 * depth)
            n_p = max(spb, int(0.5 / max(cal.get("pipelined_ms_per_sample", 1.0) * 1e-3, 1e-6)))
            e_p, st_p, g_p, r_p, f_p = timed("pipelined", 2, max(1, n_p // 2), only="seeds")
            leg_p = leg_summary("pipelined", max(1, n_p // 2), 2, e_p, st_p, g_p, r_p, f_p)
            n_q = max(spb, int(0.2 / max(cal.get("sequential_ms_per_sample", 1.5) * 1e-3, 1e-6)))
            e_q, st_q, g_q, r_q, f_q = timed("sequential", 1, n_q)
            leg_q = leg_summary("sequential", n_q, 1, e_q, st_q, g_q, r_q, f_q)
            packed_leg = {"what": "the same samples with the reads resident in HBM as the packed 2-bit stream (SYLPH_ENC_2BIT, 0.25 B/base in)",
                          "pipelined": {k_: leg_p[k_] for k_ in ("value", "ms_per_sample", "timed_region_s", "probe_batch_mean", "kernel_ms")},
                          "one_step_at_a_time": {k_: leg_q[k_] for k_ in ("value", "ms_per_sample", "timed_region_s", "kernel_ms")},
                          "table_equal_to_ascii": None}
            # the packed stream must give the very table the ASCII bytes give (first read set, both encodings, compared on the device)
            sk_a, (ka, ca, na, da) = sketch_inline(read_sets[0])
            sk_b, (kb, cb_, nb_, db_) = sketch_inline(packed_sets[0])
            packed_leg["table_equal_to_ascii"] = bool(na == nb_ and da == db_ and torch.equal(SH.device_view(ka, na, torch.int64, device), SH.device_view(kb, nb_, torch.int64, device))
                                                      and torch.equal(SH.device_view(ca, na, torch.int32, device), SH.device_view(cb_, nb_, torch.int32, device)))
            sk_a.close(); sk_b.close()
        except Exception as e:
            packed_leg = {"error": str(e)}
        active_sets[0] = read_sets
        sample_no[0] = 0
        packed_sets = None

    # ---- the same samples with the pair set behind the reference's DEFAULT cuckoo filter (--fpr 1e-4, sketch.rs:733-769; csrc/a10.hip).
    # `value` stays on the exact set (sylph's --fpr 0): the filter's hash bits are this repository's model of a crate that is not in
    # the reference tree, so only the exact set is pinned to the reference's own arithmetic.  Reported beside it.
    filter_leg = None
    if not args.no_filter_leg and not long_mode and comm is None and db_mode != "genome-py":
        try:
            def with_filter(on):
                pipe_box[0].set_option("dedup_fpr", "1e-4" if on else "0")
                filter_box[0] = on
            with_filter(True)
            runners["pipelined"](2 * depth)
            n_p = max(spb, int(0.5 / max(cal.get("pipelined_ms_per_sample", 1.0) * 1e-3, 1e-6)))
            e_p, st_p, g_p, r_p, f_p = timed("pipelined", 2, max(1, n_p // 2), only="seeds")
            leg_p = leg_summary("pipelined", max(1, n_p // 2), 2, e_p, st_p, g_p, r_p, f_p)
            n_q = max(spb, int(0.2 / max(cal.get("sequential_ms_per_sample", 1.5) * 1e-3, 1e-6)))
            e_q, st_q, g_q, r_q, f_q = timed("sequential", 1, n_q)
            leg_q = leg_summary("sequential", n_q, 1, e_q, st_q, g_q, r_q, f_q)
            filter_leg = {"what": "the same samples with sylph's default pair dedup: the pair set behind a scalable cuckoo filter, --fpr 1e-4, capacity 10^7 "
                                  "(sketch.rs:733-769); bit-exact against the oracle's model of that filter, whose hash bits are not the crate's",
                          "pipelined": {k_: leg_p[k_] for k_ in ("value", "ms_per_sample", "timed_region_s", "probe_batch_mean", "kernel_ms")},
                          "one_step_at_a_time": {k_: leg_q[k_] for k_ in ("value", "ms_per_sample", "timed_region_s", "kernel_ms")},
                          "dup_removed": int(np.mean([r[3] for r in r_p])), "dup_removed_exact_set": int(np.mean([r[3] for r in rows]))}
            if not args.no_cpu_baseline and rank == 0:
                sk_f, (kf, cf, nf, df) = sketch_inline(read_sets[0])
                hb = read_sets[0]["bases"][:read_sets[0]["n_bases"]].cpu().numpy()
                ho = read_sets[0]["rec_off"].cpu().numpy().astype(np.uint64)
                t_o = time.time()
                from oracle import oracle as O  # noqa: F811  (the checker of this leg's verify, never the thing measured)
                e = O.sketch_reads_cuckoo_model(hb, ho, c=c_reads, k=k, mode=O.MODE_AVX2_FAST if O.lib().orc_has_avx2() else O.MODE_SCALAR, fpr=1e-4)
                filter_leg["verify"] = {"what": "the whole table of the first read set vs the oracle's walk of the same filter (one CPU thread)",
                                        "table_equal": bool(nf == len(e["kmers"]) and df == e["dup_removed"]
                                                            and np.array_equal(SH.device_view(kf, nf, torch.int64, device).cpu().numpy().view(np.uint64), e["kmers"])
                                                            and np.array_equal(SH.device_view(cf, nf, torch.int32, device).cpu().numpy().view(np.uint32), e["counts"])),
                                        "entries": int(nf), "dup_removed": int(df), "oracle_seconds": round(time.time() - t_o, 2)}
                sk_f.close()
        except Exception as e:
            filter_leg = {"error": str(e)}
        try:
            with_filter(False)
        except Exception:
            pass
        sample_no[0] = 0

    comparisons = world * n_total                                             # every sample vs every genome of the DB
    parallelism = (f"{world} GPU(s), {mode}: " + (f"{n_workers} sketch worker contexts + 1 profile context per GPU, {depth} samples in flight, <= {max(spb, 8) if comm is None else spb} tables per probe launch" if mode == "pipelined" else f"one context per GPU, {spb} table(s) per probe launch") + "; database " +
                   ("sharded by GENOME over the GPUs (contiguous ranges balanced by k-mer count, an unsharded index each): every table all-gathered to "
                    "every rank and probed there, ONE all-gather of contain_count[S, G/W] + one of the coverage values (torch.distributed on "
                    f"{'device tensors (RCCL)' if (dist is not None and dist.get_backend() == 'nccl') else 'host copies'}; sylph_amd/shard.py)"
                    if db_mode == "genome-py" else
                    (f"sharded by GENOME over {world} GPUs inside the library (sylph_db_upload_genome_shard: contiguous genome ranges balanced by k-mer count, global ids; per probe "
                     f"batch every table travels whole to every shard, hits all-to-all to the owners, two tiny all-gathers of sizes; {comm_kind})" if db_mode == "genome" else
                     f"sharded by k-mer range over {world} GPUs (per probe batch: table slices all-to-all, hits all-to-all to the owners, two tiny all-gathers of sizes; "
                     f"{comm_kind})")
                    if comm is not None else ("replicated on every GPU (no data-path collective)" if world > 1 else "on the one GPU")))
    out = {
        "metric": "read Gbp/s sketched + genome-comparisons/s profiled, 1 Gbp vs GTDB-R220",
        "value": main_leg["value"], "unit": "Gbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": DESCR[wl], "samples_per_gpu_per_step": sps, "samples_per_probe_batch": spb, "mode": mode,
                   "sketch_workers_per_gpu": n_workers if mode == "pipelined" else 0, "samples_in_flight": depth if mode == "pipelined" else 1,
                   **({"fallbacks": fallbacks} if fallbacks else {}), "reads_per_sample_gbp": round(n_bases / 1e9, 4),
                   "distinct_read_sets_rotated": n_sets, "genomes": n_total, "db_kmers_per_shard": dbstats["shard_kmers"],
                   "dedup": ("none applies (reads > 400 bp, sketch.rs:922-927)" if long_mode else
                             f"the pair set behind the model of sylph's cuckoo filter, --fpr {args.main_dedup_fpr} (profiling run)" if args.main_dedup_fpr else "exact (--fpr 0 semantics)"),
                   "seed_mode": "avx2_compat", "parallelism": parallelism,
                   "db_mode": db_mode if world > 1 else "one GPU (whole index)", **({"shard_reduce": args.shard_reduce} if comm is not None else {}),
                   "inputs": "reads + database resident in HBM before the timed region",
                   "rng": "splitmix64 counter streams (synth.py: every random number = one word of a stream addressed by its index, integer arithmetic "
                          "only; host float tables for the abundances / decoy lengths), read set seed 20250711 + 1000003*(rank+1) + 7919*set"},
        "mode": mode, "value_is": f"{mode}: all bases of the {args.steps} timed steps / their wall time (max over ranks)",
        "timed_region_s": main_leg["timed_region_s"], "ms_per_sample": main_leg["ms_per_sample"],
        "step_ms": main_leg["step_ms"], "sample_interval_ms": main_leg["sample_interval_ms"],
        "calibration": cal,
        "sketch_gbp_per_s": main_leg["sketch_gbp_per_s"], "genome_comparisons_per_s": main_leg["genome_comparisons_per_s"],
        "genome_comparisons_per_s_whole_step": round(comparisons / (elapsed / (sps * args.steps)), 1),
        "sketch_ms": main_leg["sketch_ms"], "profile_ms": main_leg["profile_ms"], "probe_batch_mean": main_leg["probe_batch_mean"],
        "sample_table_entries": int(np.mean([r[2] for r in rows])), "dup_removed": int(np.mean([r[3] for r in rows])),
        # per family: (average ms per launch group, launch groups).  The timed region times the dominant kernel only; the other families
        # are from `kernel_ms_pass` (every family timed, a separate shorter pass of the same mode)
        "kernel_ms": {**(main_leg["kernel_ms_pass"]["kernel_ms"] if "kernel_ms_pass" in main_leg else {}), **main_leg["kernel_ms"]},
        "timers_in_timed_region": "every family" if args.all_kernel_timers else ("none" if args.no_kernel_timers else "the dominant kernel's family (seeds) only"),
        **({"kernel_ms_pass": main_leg["kernel_ms_pass"]} if "kernel_ms_pass" in main_leg else {}),
        **({"exchange": main_leg["exchange"]} if "exchange" in main_leg else {}),
        "setup": dbstats,
        **({"resident_2bit": packed_leg} if packed_leg is not None else {}),
        **({"default_pair_dedup": filter_leg} if filter_leg is not None else {}),
    }
    # what `sylph sketch -1 -2` / `profile -1 -2` do by DEFAULT is the filter rule (cmdline.rs:77, contain.rs:591): its rate beside `value`
    if filter_leg is not None and "pipelined" in filter_leg:
        out["value_default_flags"] = filter_leg["pipelined" if mode == "pipelined" else "one_step_at_a_time"]["value"]
    legs = {mode: (fam, rows)}
    if second is not None:
        o_mode, per2, e2, st2, g2, r2, f2 = second
        steps2 = len(st2)
        leg2 = leg_summary(o_mode, per2, steps2, e2, st2, g2, r2, f2)
        out["one_step_at_a_time" if o_mode == "sequential" else "pipelined"] = leg2
        legs[o_mode] = (f2, r2)
    out["one_step_at_a_time" if mode == "sequential" else "pipelined"] = {k_: main_leg[k_] for k_ in main_leg if k_ not in ("kernel_ms", "kernel_ms_pass")} | {"same_as": "the top-level fields"}

    # roofline of the dominant kernel (seeds): algorithmic bytes per launch = 1 B/base + 8 B/record offset + 8 B/seed
    # occurrence out, over the HIP-event duration of that launch.
    fp = csrc_fingerprint()
    meta = {}
    try:
        meta = json.load(open(os.path.join(ROOT, "profiles", "seeds_traffic.json")))
    except Exception:
        pass
    meta_ok = meta.get("csrc_sha") == fp     # PMC figures measured on exactly these kernel sources
    tsrc = f"profiles/seeds_traffic.json@{meta.get('head', '?')[:12]} (rocprofv3 --pmc passes of tools/r05_profile.sh; csrc {meta.get('csrc_sha', '?')})"
    seeds_ms, seeds_launches = fam["seeds"]
    if seeds_launches:
        n_rec = float(np.mean([r["n_records"] for r in read_sets]))
        n_occ = float(np.mean(last["occ"]))
        launches_per_sample = max(1, round(seeds_launches / (args.steps * sps)))
        # a sample > 2^32 bases is pushed in batches (long reads: every launch is a batch of its own, bytes and time per batch); a short-read
        # sample in a pipeline is ONE batch whose last tenth of blocks is a second launch of the same kernel behind the turn's event
        # (reads_tail_pct, round 6): its two launches are one "launch" here — all the sample's bytes, the SUM of the two durations
        groups = launches_per_sample if long_mode else 1
        alg_bytes = (n_bases + 8 * n_rec + 8 * n_occ) / groups
        avg_ms = seeds_ms / seeds_launches * (launches_per_sample / groups)
        seeds_launches = seeds_launches * groups / launches_per_sample
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = meta.get("hbm_bytes_per_launch") if (meta_ok and wl in ("c2", "c3", "c4")) else None
        if meta_ok and long_mode:
            traffic = meta.get("position_kernel_hbm_bytes_per_launch")   # the C5 position kernel's own PMC pass (tools/r04_profile.sh)
        ipk = (meta.get("valu_per_kmer_position_kernel") if long_mode else meta.get("valu_per_kmer")) if meta_ok else None
        hashed = (n_bases if long_mode else max(0.0, n_bases - n_rec * (k - 1))) / groups
        out["roofline"] = {"bound": "hbm", "kernel": "seeds_slots_kernel<31,1>" if long_mode else "reads_kernel<31,1>", "achieved": round(achieved, 1),
                           "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                           "traffic_source": tsrc if traffic is not None else f"none: no PMC figures for these kernel sources (csrc {fp}; profiles/seeds_traffic.json holds {meta.get('csrc_sha', 'nothing')})",
                           "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(avg_ms, 4), "launches": int(seeds_launches),
                           "note": "integer-VALU issue bound (SQ counters in profiles/)" +
                                   (f"; pipelined: launch durations include time shared with the other streams' kernels (alone on the GPU: alone_on_gpu)" if mode == "pipelined" else "")}
        if not long_mode and launches_per_sample > 1:
            # Since round 6 a pipeline launches the last tenth of a sample's blocks behind its turn's event, so consecutive samples' seeding
            # kernels OVERLAP on purpose (+0.9 % samples per second): each runs longer than it would behind the other, and the sum of a sample's
            # two launch durations exceeds the time between two samples.  What the chip does with this kernel's bytes per second of wall
            # clock is bytes per sample / sample period:
            period_ms = elapsed / (args.steps * sps) * 1e3
            out["roofline"]["launches_per_sample"] = int(launches_per_sample)
            out["roofline"]["overlapping_launches"] = {
                "sum_of_a_samples_launch_ms": round(avg_ms, 4), "sample_period_ms": round(period_ms, 4),
                "achieved_per_wall_time": round(alg_bytes / (period_ms * 1e-3) / 1e9, 1), "frac_per_wall_time": round(alg_bytes / (period_ms * 1e-3) / 1e9 / 8000.0, 4),
                "note": "avg_launch_ms is the SUM of a sample's two launches of this kernel (head + tail, reads_tail_pct = 10); consecutive samples' kernels overlap, "
                        "so `frac` (bytes / that sum) fell from 0.148 to ~0.12 while the sample rate rose: frac_per_wall_time = bytes per sample / sample period"}
        if ipk:
            # secondary ceiling: VALU issue = 256 CU x 4 SIMD x 16 lanes x 2.4 GHz lane-ops/s, i.e. 4 cycles per
            # wave-instruction; plain VOP2 integer ops issue faster than that on this chip (profiles/r02_valu_rates.txt),
            # so a launch alone on the GPU can come out a few per cent above 1.0
            out["roofline"]["valu_ceiling"] = {"instr_per_kmer": ipk, "kmers_per_launch": int(hashed),
                                               "min_ms": round(ipk * hashed / (256 * 4 * 16 * 2.4e9) * 1e3, 4),
                                               "frac": round(ipk * hashed / (256 * 4 * 16 * 2.4e9) * 1e3 / avg_ms, 3)}
        if "sequential" in legs and legs["sequential"][0]["seeds"][1]:
            f1 = legs["sequential"][0]
            a1 = f1["seeds"][0] / f1["seeds"][1]
            out["roofline"]["alone_on_gpu"] = {"avg_launch_ms": round(a1, 4), "achieved": round(alg_bytes / (a1 * 1e-3) / 1e9, 1),
                                               "frac": round(alg_bytes / (a1 * 1e-3) / 1e9 / 8000.0, 4), "launches": int(f1["seeds"][1])}
    # roofline of the filter dedup (the reference's default for pairs; csrc/a10.hip): one "launch" = the six dispatches of the partitioned
    # pass of one sample (operation words, one partition level by class range, in-LDS resolution), timed alone on the GPU by the library's
    # HIP events.  Algorithmic bytes: the 32 B occurrence record of every seed occurrence in (k-mer, two markers, record id: what the
    # filter's items are made of); the marks go back into ~3 % of the records in place.  `traffic`: the four kernels' PMC counters.
    if filter_leg is not None and filter_leg.get("one_step_at_a_time", {}).get("kernel_ms", {}).get("a10"):
        a_ms, a_n = filter_leg["one_step_at_a_time"]["kernel_ms"]["a10"]
        n_occ_f = float(np.mean(last["occ"])) if last.get("occ") else 0.0
        alg = 32.0 * n_occ_f
        a_traffic = meta.get("a10_hbm_bytes_per_sample") if meta_ok else None
        out["roofline_a10"] = {"bound": "hbm", "kernel": "a10_ops_tile_kernel (operation words + their histogram) + part_scan/scatter (by class range) + a10_range_kernel",
                               "achieved": round(alg / (a_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (a_ms * 1e-3) / 1e9 / 8000.0, 4),
                               "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": a_ms, "launches": int(a_n),
                               "traffic": a_traffic, "traffic_over_algorithmic": round(a_traffic / alg, 2) if (a_traffic and alg) else None,
                               "traffic_frac_of_peak": round(a_traffic / (a_ms * 1e-3) / 1e9 / 8000.0, 4) if a_traffic else None,
                               "traffic_source": tsrc if a_traffic else "none: no PMC figures for these kernel sources",
                               "note": "8 B operation words through ONE partition level (round 5: two), ranges of ~8,000 words resolved through an LDS bit table by "
                                       "256-thread workgroups sized to fit beside the next samples' seeding kernels (profiles/r06_ab_a10.txt); round 4's table "
                                       "of 2 atomics + 1 load per operation: 0.92 ms.  The traffic is several times the algorithmic bytes by construction, all of it coalesced"}
    # roofline of the profile half: probe_kernel, one launch per probe batch over all tables it probes.  Inverted-index formulation
    # (SURVEY 8d): B = N_s * (8 + 4 table in + 64 one index line per probe) + 8 * hits out.
    fam_mix, rows_mix = (fam_pass["fam"], fam_pass["rows"]) if fam_pass is not None else (fam, rows)
    probe_ms, probe_launches = fam_mix["probe"]
    if probe_launches:
        def probe_roof(fm, rws):
            pms, pl = fm["probe"]
            batch = len(rws) / pl                                      # tables per launch
            probes = float(np.mean([r[2] for r in rws])) * batch
            hits = float(np.mean([r[4] for r in rws])) * batch
            # SURVEY 8d, inverted-index form: N_s * (8 table in + 16 probe) + 4 * postings hit + 4 * G * S counts out — what
            # `achieved` / `frac` are quoted on.  The access granule is a whole 64 B index line per probe (and the kernel writes
            # 8 B per hit): `line_granule` keeps that accounting beside it, it is NOT the roofline figure.
            alg = probes * 24 + 4 * hits + 4 * n_total * batch
            granule = probes * (12 + 64) + 8 * hits
            avg = pms / pl
            return {"achieved": round(alg / (avg * 1e-3) / 1e9, 1), "frac": round(alg / (avg * 1e-3) / 1e9 / 8000.0, 4),
                    "algorithmic_bytes_per_launch": int(alg), "algorithmic_bytes_formula": "N_s*(8+16) + 4*hits + 4*G*S (SURVEY 8d, inverted index)",
                    "avg_launch_ms": round(avg, 4), "launches": int(pl),
                    "tables_per_launch": round(batch, 2), "probes_per_launch": int(probes), "hits_per_launch": int(hits),
                    "line_granule": {"bytes_per_launch": int(granule), "achieved": round(granule / (avg * 1e-3) / 1e9, 1),
                                     "frac": round(granule / (avg * 1e-3) / 1e9 / 8000.0, 4), "over_algorithmic": round(granule / alg, 2),
                                     "what": "one 64 B index line + 12 B table entry per probe, 8 B per hit written"}}
        pr = probe_roof(fam_mix, rows_mix)
        per_probe = meta.get("probe_hbm_bytes_per_probe") if (meta_ok and wl in ("c3", "c4", "c3r")) else None
        out["roofline_profile"] = {"bound": "hbm", "kernel": "probe_kernel", "peak": 8000.0, "unit": "GB/s", **pr,
                                   "traffic": int(per_probe * pr["probes_per_launch"]) if per_probe else None,
                                   "traffic_over_algorithmic": round(per_probe * pr["probes_per_launch"] / max(1, pr["algorithmic_bytes_per_launch"]), 2) if per_probe else None,
                                   "traffic_source": tsrc if per_probe else "none: no PMC figures for these kernel sources",
                                   "note": "random 64 B line reads, latency-bound; one index line per probe is the access granule"}
        if "sequential" in legs and legs["sequential"][0]["probe"][1]:
            out["roofline_profile"]["alone_on_gpu"] = probe_roof(*legs["sequential"])
    if not args.no_verify and last.get("res") is not None:
        try:
            G = n_total
            cc, off, covs = last["res"]
            out["verify"] = verify_against_oracle(ctx, (cc[:G], off[:G + 1], covs), G, last["table"], verify_set, device)
        except Exception as e:
            out["verify"] = {"genomes_checked": 0, "mismatches": None, "error": str(e)}
    for sk in last.get("sessions", []):
        sk.close()
    pipe_box[0].close()
    if args.sweep and comm is None:
        # tuning: other pipeline shapes / options on the same resident workload, ~0.8 s each, two rounds so that drift shows
        spec = json.load(open(args.sweep)) if os.path.exists(args.sweep) else json.loads(args.sweep)
        sweep = []
        for rnd in range(2):
            for cfg in spec:
                w_, d_ = int(cfg.get("workers", 2)), int(cfg.get("depth", 4))
                p = S.Pipeline(db, c=c_reads, k=k, paired=not long_mode, n_workers=w_, depth=d_, max_batch=int(cfg.get("max_batch", 8)))
                for k_, v_ in cfg.get("options", {}).items():
                    p.set_option(k_, str(v_))
                pipe_box[0] = p
                depth_saved, depth = depth, d_
                try:
                    run_pipelined(3 * d_)
                    n_s = int(cfg.get("samples", 800))
                    rows_s = []
                    torch.cuda.synchronize()
                    t_s = time.perf_counter()
                    run_pipelined(n_s, None, rows_s)
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t_s) / n_s
                    rec = {"name": cfg.get("name", "?"), "round": rnd, "ms_per_sample": round(dt * 1e3, 4), "gbp_per_s": round(n_bases / 1e9 / dt, 1),
                           "probe_batch_mean": round(float(np.mean([r[5] for r in rows_s])), 2)}
                except Exception as e:
                    rec = {"name": cfg.get("name", "?"), "round": rnd, "error": str(e)}
                depth = depth_saved
                log(f"[sweep] {json.dumps(rec)}")
                sweep.append(rec)
                p.close()
        out["sweep"] = sweep
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import oracle as O  # noqa: F401  (cpu_baseline leg only)
            rs = read_sets[0]
            seq_k, seq_off = verify_set[3]
            n_decoy = n_total - (len(seq_off) - 1)
            try:
                avail = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) * 1024
            except Exception:
                avail = 0
            need = 8 * dbstats["db_kmers_total"] * 2                      # the genome-major database + the oracle's coverage array
            whole = avail > need + (16 << 30) and not args.cpu_baseline_bounded
            G_dec = n_decoy if whole else min(n_decoy, 16000)
            # the oracle needs the genome-major layout on the host: the sequence-backed sketches are there, the decoys are regenerated
            if G_dec > 0:
                dk, doff = synth.decoy_sketches(n_decoy, c=c, device=device, seed=args.seed + 7)
                doff_h = doff.cpu().numpy().astype(np.uint64)[:G_dec + 1]
                dk_h = dk[:int(doff_h[-1])].cpu().numpy().view(np.uint64)
                del dk, doff
            else:
                doff_h, dk_h = np.zeros(1, np.uint64), np.zeros(0, np.uint64)
            if whole:
                db_host = (np.concatenate([seq_k, dk_h]), np.concatenate([seq_off.astype(np.uint64), doff_h[1:] + np.uint64(seq_off[-1])]))
            else:
                db_host = (dk_h, doff_h)
            del dk_h
            cb = cpu_baseline(rs["bases"], rs["rec_off"], rs["n_records"], not long_mode, c_reads, k, db_host, whole)
            t_cpu = n_bases / 1e9 / cb["sketch_gbp_per_s"] + n_total / cb["comparisons_per_s"]
            out["cpu_baseline"] = {"value": round(n_bases / 1e9 / t_cpu, 4), "unit": "Gbp/s", "cores": cb["probe_cores"], "host_hardware_threads": os.cpu_count(), "kind": "port",
                                   "sample": cb["sample"], "sketch_gbp_per_s": round(cb["sketch_gbp_per_s"], 4),
                                   "sketch_cores": 1, "genome_comparisons_per_s": round(cb["comparisons_per_s"], 1),
                                   "sketch_s": round(cb["sketch_s"], 3), "probe_s": round(cb["probe_s"], 3), "extrapolated": not whole,
                                   "note": "C++ restatement of the reference CPU path (oracle/); the reference sketches one sample on one thread and probes genomes on all threads"}
            if whole and not args.no_verify:
                # full-size parity inside the run (untimed): the GPU's table of the SAME read set against the oracle's, then every
                # genome's containment count and every genome-with-hits' sorted coverage vector against the oracle's probe
                sk_g, (dk_g, dc_g, nt_g, dup_g) = sketch_inline(rs)
                gk = SH.device_view(dk_g, nt_g, torch.int64, device).cpu().numpy().view(np.uint64)
                gc = SH.device_view(dc_g, nt_g, torch.int32, device).cpu().numpy().view(np.uint32)
                res = db.contain_batch([(dk_g, dc_g, nt_g)], device_ptrs=True)
                cc_g, off_g, cov_g = (np.array(x) for x in res)
                sk_g.close()
                tab = cb["table"]
                table_ok = bool(np.array_equal(gk, tab["kmers"]) and np.array_equal(gc, tab["counts"]) and int(dup_g) == int(tab["dup_removed"]))
                ecc, ecov, ego = cb["contain_count"], cb["cov"], cb["genome_off"]
                bad = int((cc_g[:n_total] != ecc).sum())
                hit_g = np.nonzero(ecc)[0]
                for g in hit_g:
                    got = np.asarray(cov_g[int(off_g[g]):int(off_g[g + 1])]).astype(np.uint32)
                    exp = np.sort(ecov[int(ego[g]):int(ego[g]) + int(ecc[g])])
                    if len(got) != len(exp) or not np.array_equal(got, exp):
                        bad += 1
                if "verify" in out:
                    out["verify_subset"] = out["verify"]
                out["verify"] = {"genomes_checked": int(n_total), "genomes_with_hits": int(len(hit_g)), "hits_checked": int(ecc.sum()),
                                 "mismatches": int(bad), "sample_table_entries": int(nt_g), "sample_table_equal": table_ok,
                                 "what": "the whole sample at full size: the GPU's (k-mer, count) table and duplicate count vs the oracle's sketch of the "
                                         f"same {n_bases / 1e9:.0f} Gbp read set; contain_count of EVERY genome of the database and the sorted coverage vector of every "
                                         "genome with hits vs the oracle's probe of that table"}
                if not table_ok:
                    out["verify"]["mismatches"] = int(bad) + 1
        except Exception as e:  # the baseline leg must never sink the GPU measurement
            out["cpu_baseline"] = {"value": None, "unit": "Gbp/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    if rank == 0 and world == 1 and not args.no_files_leg and not args.no_h2d and not long_mode and wl in ("c2", "c3", "c4"):
        try:
            out["end_to_end_from_files"] = end_to_end_from_files(read_sets[0]["bases"], n_pairs, read_len, args.files_leg_pairs)
        except Exception as e:
            out["end_to_end_from_files"] = {"error": str(e)[:400]}
    if rank == 0:
        # every rate a reader needs to compare, side by side at the top level: GPU with the inputs resident (= `value`, and
        # with the reference's default flags), fed over PCIe, from files through the product's own command; the CPU path on the same inputs
        e2e = out.get("end_to_end_from_files", {}) if isinstance(out.get("end_to_end_from_files"), dict) else {}
        cbf = e2e.get("cpu_baseline_from_files", {}) if isinstance(e2e.get("cpu_baseline_from_files"), dict) else {}
        h2d = out.get("value_h2d_inclusive", {}) if isinstance(out.get("value_h2d_inclusive"), dict) else {}

        def g(dct, *ks):
            for k_ in ks:
                dct = dct.get(k_) if isinstance(dct, dict) else None
            return dct
        out["rates_gbp_per_s"] = {
            "gpu_inputs_resident_exact_dedup": out.get("value"), "gpu_inputs_resident_default_flags": out.get("value_default_flags"),
            "gpu_fed_over_pcie_ascii": g(h2d, "ascii", "gbp_per_s"), "gpu_fed_over_pcie_2bit": g(h2d, "2bit", "gbp_per_s"),
            "gpu_from_files_plain_1_sample_command": g(e2e, "plain_one_sample", "command_gbp_per_s"),
            "gpu_from_files_plain_4_samples_command": g(e2e, "plain_four_samples_one_command", "command_gbp_per_s"),
            "gpu_from_files_gz_1_sample_command": g(e2e, "gz_one_sample", "command_gbp_per_s"),
            "gpu_from_files_gz_4_samples_command": g(e2e, "gz_four_samples_one_command", "command_gbp_per_s"),
            "cpu_resident_whole_job": g(out, "cpu_baseline", "value"), "cpu_resident_sketch_one_thread": g(out, "cpu_baseline", "sketch_gbp_per_s"),
            "cpu_from_files_plain": {k_: g(cbf, k_, "gbp_per_s") for k_ in ("plain_1_samples", "plain_4_samples", "plain_16_samples")},
            "cpu_from_files_gz": {k_: g(cbf, k_, "gbp_per_s") for k_ in ("gz_1_samples", "gz_4_samples", "gz_16_samples")},
            "cpu_cores_usable": cbf.get("cores_usable"),
            "note": "command = whole `sylph-hip sketch` process (start-up, GPU bring-up, feed, kernels, .sylsp written); cpu_from_files = the oracle "
                    "(kind: port) on the same files, one thread per sample as the reference runs samples; sketch stage only on both sides",
        }
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
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
