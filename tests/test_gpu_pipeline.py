"""GPU tests of sylph_pipeline_* (csrc/pipeline.hip): a stream of samples through sketch workers + the profile thread must give,
sample by sample and in submission order, exactly what the one-sample-at-a-time entry points and the CPU oracle give — whatever
the number of workers, the depth and the probe batching."""
import numpy as np
import pytest

import sylph_amd as S
from oracle import oracle as O
from sylph_amd.binding import MEM_HOST

from .helpers import concat, random_seq
from .test_gpu_parity import make_reads

pytestmark = pytest.mark.gpu


def small_world(seed, n_genomes=5, glen=40000, c=50):
    rng = np.random.default_rng(seed)
    genomes = [random_seq(rng, glen) for _ in range(n_genomes)]
    gk = [O.sketch_genome(g, np.array([0, len(g)], dtype=np.uint64), c=c)["genome_kmers"] for g in genomes]
    goff = np.zeros(len(gk) + 1, dtype=np.uint64)
    goff[1:] = np.cumsum([len(x) for x in gk])
    return rng, genomes, np.concatenate(gk), goff


def sample_reads(rng, genomes, which, n, paired=True):
    recs = []
    for g in which:
        recs += make_reads(rng, genomes[g], n, 150, paired=paired, dup_frac=0.05, ragged=False)
    return concat(recs)


def check_result(r, e, db_k, goff, mk=50.0):
    assert r["n_table"] == len(e["kmers"]) and r["dup_removed"] == e["dup_removed"]
    ecc, ecov, _ = O.contain(e["kmers"], e["counts"], db_k, goff, min_number_kmers=mk)
    assert np.array_equal(r["contain_count"], ecc)
    off, covs = r["cov_off"], r["covs"]
    for g in range(len(goff) - 1):
        assert np.array_equal(np.asarray(covs[int(off[g]):int(off[g + 1])]).astype(np.uint32), np.sort(ecov[g])), g


@pytest.mark.parametrize("workers,depth,max_batch", [(1, 1, 1), (2, 4, 8), (3, 6, 2)])
def test_pipeline_matches_oracle_in_submission_order(ctx, workers, depth, max_batch):
    import torch
    rng, genomes, db_k, goff = small_world(5)
    db = S.Database(ctx, db_k, goff)
    samples = []
    for i in range(9):
        b, off = sample_reads(rng, genomes, [i % 5, (i * 2 + 1) % 5][: 1 + i % 2], 400 + 150 * (i % 3))
        if i == 4:
            b, off = np.zeros(0, np.uint8), np.zeros(1, np.uint64)           # an empty sample in the middle of the stream
        samples.append((b, off, O.sketch_reads(b, off, c=50, paired=True)))
    dev = [(torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()) for b, off, _ in samples]
    torch.cuda.synchronize()
    p = S.Pipeline(db, c=50, paired=True, n_workers=workers, depth=depth, max_batch=max_batch, want_table=True)
    submitted = done = 0
    while done < len(samples):
        while submitted < len(samples) and p.outstanding < depth:
            tb, toff = dev[submitted]
            b, off, _ = samples[submitted]
            ok = p.submit_device([(tb.data_ptr(), toff.data_ptr(), len(off) - 1, int(off[-1]))], tag=1000 + submitted)
            assert ok
            submitted += 1
        if submitted < len(samples):      # depth reached: one more must be refused, not block
            tb, toff = dev[submitted]
            assert p.submit_device([(tb.data_ptr(), toff.data_ptr(), len(samples[submitted][1]) - 1, int(samples[submitted][1][-1]))]) is False
        r = p.next()
        assert r["tag"] == 1000 + done                                        # submission order
        e = samples[done][2]
        check_result(r, e, db_k, goff)
        assert np.array_equal(r["kmers"], e["kmers"]) and np.array_equal(r["counts"], e["counts"])
        assert 1 <= r["probe_batch"] <= max_batch
        t = r["t"]
        assert t[0] <= t[1] <= t[2] <= t[3] <= t[4]
        done += 1
    with pytest.raises(S.SylphHipError):
        p.next()                                                              # nothing outstanding
    p.close()
    db.close()


@pytest.mark.parametrize("tail_pct", [0, 10, 50])
def test_pipeline_seeding_turns_and_the_tail_launch(ctx, tail_pct):
    """One seeding kernel at a time (common.h SeedTurn): a worker's stream waits for the previous sample's seeding kernel right in front
    of its own, and with "reads_tail_pct" the last share of a sample's blocks is a launch of its own behind the turn's event (the next
    sample's kernel starts while it runs).  Samples of ~100 blocks of reads through three workers, exact set and default pair dedup,
    the old placement of the wait too ("serialize_outside"): every table and result equal to the oracle's, in submission order."""
    import torch
    rng, genomes, db_k, goff = small_world(11)
    db = S.Database(ctx, db_k, goff)
    samples = []
    for i in range(6):
        b, off = sample_reads(rng, genomes, [i % 5, (i + 2) % 5], 6000 + 500 * (i % 3))
        samples.append((b, off))
    dev = [(torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()) for b, off in samples]
    torch.cuda.synchronize()
    for fpr, outside in ((None, 0), (1e-4, 0), (None, 1)):
        exp = [O.sketch_reads(b, off, c=50, paired=True) if fpr is None else O.sketch_reads_cuckoo_model(b, off, c=50, fpr=fpr) for b, off in samples]
        p = S.Pipeline(db, c=50, paired=True, n_workers=3, depth=6, max_batch=4, want_table=True)
        p.set_option("reads_tail_pct", tail_pct)
        p.set_option("serialize_outside", outside)
        if fpr is not None:
            p.set_option("dedup_fpr", fpr)
        for rnd in range(2):
            for i, (tb, toff) in enumerate(dev):
                assert p.submit_device([(tb.data_ptr(), toff.data_ptr(), len(samples[i][1]) - 1, int(samples[i][1][-1]))], tag=i)
            for i in range(len(dev)):
                r = p.next()
                assert r["tag"] == i
                check_result(r, exp[i], db_k, goff)
                assert np.array_equal(r["kmers"], exp[i]["kmers"]) and np.array_equal(r["counts"], exp[i]["counts"])
        p.close()
    db.close()


def test_pipeline_with_the_default_pair_dedup(ctx):
    """sylph_pipeline_set_option("dedup_fpr"): every session the pipeline opens from then on deduplicates its pairs behind the
    cuckoo filter (sketch.rs:733-769; csrc/a10.hip) — the samples' tables, duplicate counts and containment rows must be the oracle's
    model of that filter's; setting it back to 0 returns to the exact set.  Device batches (the deferred seeding verdict is
    resolved by the filter pass) and host batches."""
    import torch
    rng, genomes, db_k, goff = small_world(11)
    db = S.Database(ctx, db_k, goff)
    fpr, cap = 0.05, 2500                                        # leaky and small: it grows and reports false positives
    samples = []
    for i in range(5):
        b, off = sample_reads(rng, genomes, [i % 5], 500 + 100 * i)
        samples.append((b, off, O.sketch_reads_cuckoo_model(b, off, c=50, fpr=fpr, initial_capacity=cap), O.sketch_reads(b, off, c=50, paired=True)))
    assert any(f["dup_removed"] != x["dup_removed"] for _, _, f, x in samples)
    dev = [(torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()) for b, off, _, _ in samples]
    torch.cuda.synchronize()
    p = S.Pipeline(db, c=50, paired=True, n_workers=2, depth=4, max_batch=4, want_table=True)
    p.set_option("dedup_fpr", fpr)
    p.set_option("dedup_capacity", cap)
    for i, (b, off, _, _) in enumerate(samples):
        if i % 2 == 0:
            assert p.submit_device([(dev[i][0].data_ptr(), dev[i][1].data_ptr(), len(off) - 1, int(off[-1]))], tag=i)
        else:
            assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=i, mem=MEM_HOST)
        if p.outstanding == 4 or i + 1 == len(samples):
            while p.outstanding:
                r = p.next()
                e = samples[r["tag"]][2]
                check_result(r, e, db_k, goff)
                assert np.array_equal(r["kmers"], e["kmers"]) and np.array_equal(r["counts"], e["counts"])
    p.set_option("dedup_fpr", 0)
    b, off, _, x = samples[0]
    assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=77, mem=MEM_HOST)
    r = p.next()
    check_result(r, x, db_k, goff)
    assert np.array_equal(r["counts"], x["counts"])
    p.close()
    db.close()


def test_pipeline_host_batches_sessions_and_errors(ctx):
    """Host-memory batches (several per sample), adopted sessions, and a sample that fails (odd number of records in a paired
    session) — the failure belongs to that sample only, the stream goes on."""
    rng, genomes, db_k, goff = small_world(6)
    db = S.Database(ctx, db_k, goff)
    p = S.Pipeline(db, c=50, paired=True, n_workers=2, depth=4, max_batch=4)
    b1, o1 = sample_reads(rng, genomes, [0], 500)
    b2, o2 = sample_reads(rng, genomes, [1, 2], 300)
    e1, e2 = O.sketch_reads(b1, o1, c=50, paired=True), O.sketch_reads(b2, o2, c=50, paired=True)
    # sample 0: two host batches (cut at a pair boundary)
    cut = 400
    oa = o1[:cut + 1].copy()
    ob = (o1[cut:] - o1[cut]).copy()
    ba, bb = b1[:int(o1[cut])].copy(), b1[int(o1[cut]):].copy()
    assert p.submit_device([(ba.ctypes.data, oa.ctypes.data, cut, int(oa[-1])), (bb.ctypes.data, ob.ctypes.data, len(ob) - 1, int(ob[-1]))],
                           tag=1, mem=MEM_HOST)
    # sample 1: a paired batch with an odd record count -> SYLPH_ERR_INVALID for this sample
    assert p.submit_device([(b2.ctypes.data, o2.ctypes.data, 3, int(o2[3]))], tag=2, mem=MEM_HOST)
    # sample 2: a session the caller pushed into
    sk = S.ReadSketcher(ctx, c=50, paired=True)
    sk.push(b2, o2)
    assert p.submit_session(sk, tag=3)
    r = p.next()
    assert r["tag"] == 1
    check_result(r, e1, db_k, goff)
    with pytest.raises(S.SylphHipError) as ei:
        p.next()
    assert "even number" in str(ei.value)
    r = p.next()
    assert r["tag"] == 3
    check_result(r, e2, db_k, goff)
    # kernel timers summed over the contexts
    p.profile(True)
    assert p.submit_device([(b1.ctypes.data, o1.ctypes.data, len(o1) - 1, int(o1[-1]))], tag=4, mem=MEM_HOST)
    check_result(p.next(), e1, db_k, goff)
    ms, n = p.kernel_stats("probe")
    assert n >= 1 and ms > 0
    ms, n = p.kernel_stats("seeds")
    assert n >= 1
    p.profile(False)
    # destroy with samples outstanding: they are finished first (their input memory is still ours)
    assert p.submit_device([(b1.ctypes.data, o1.ctypes.data, len(o1) - 1, int(o1[-1]))], tag=5, mem=MEM_HOST)
    p.close()
    db.close()


@pytest.mark.timeout(300)
def test_pipeline_sharded_one_rank(ctx):
    """The sharded flavour (fixed batches + flush) over a one-rank RCCL communicator equals the unsharded pipeline."""
    rng, genomes, db_k, goff = small_world(7)
    comm = S.Comm(0, 1, ctx=ctx, rccl_id=S.Comm.rccl_unique_id())
    bounds = S.shard_bounds(int(db_k.max()), 1)
    db = S.Database(ctx, db_k, goff, shard=(bounds, 1, 0))
    p = S.Pipeline(db, c=50, paired=True, n_workers=2, depth=4, max_batch=3, comm=comm)
    data = [sample_reads(rng, genomes, [i % 5], 300) for i in range(5)]
    for i, (b, off) in enumerate(data[:4]):
        assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=i, mem=MEM_HOST)
    for i in range(3):                     # the first full batch of three
        r = p.next()
        assert r["tag"] == i and r["probe_batch"] == 3
        check_result(r, O.sketch_reads(*data[i], c=50, paired=True), db_k, goff)
    b, off = data[4]
    assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=4, mem=MEM_HOST)
    p.flush()                              # the remaining two go in a partial batch
    for i in (3, 4):
        r = p.next()
        assert r["tag"] == i and r["probe_batch"] == 2
        check_result(r, O.sketch_reads(*data[i], c=50, paired=True), db_k, goff)
    # closing with fewer samples outstanding than a batch holds, never flushed: close flushes them itself (it used to wait for
    # a full batch that could not come any more)
    b, off = data[0]
    assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=9, mem=MEM_HOST)
    p.close(); db.close(); comm.close()


@pytest.mark.parametrize("db_mode", ["default", "shard", "genome"])
def test_bench_two_ranks_on_one_gpu_small_workload(db_mode):
    """`bench.py --gpus 2` end to end on whatever the box has (one GPU: the two ranks share it): the N > 1 code path of the bench —
    self-launch, agreement on mode and step size across ranks, ONE line with n_gpus = 2 whose verify leg matches the oracle — in its
    three database modes: the default (round 5: every rank a replica of the whole index, no data-path collective), `--db-mode shard`
    (k-mer ranges) and `--db-mode genome` (north_star's cut, inside the library since round 5), the last two with the library's
    exchange through torch.distributed callbacks, the pipeline's fixed probe batches with flush, and the exchange's time and bytes in
    the line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT={"default": "29577", "shard": "29578", "genome": "29579"}[db_mode])
    extra = [] if db_mode == "default" else ["--db-mode", db_mode]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "small", "--steps", "3", "--warmup", "1",
                        "--min-seconds", "0.3", "--no-cpu-baseline", "--no-h2d"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["verify"]["mismatches"] == 0 and d["verify"]["genomes_checked"] > 0
    par = d["config"]["parallelism"]
    if db_mode == "default":
        assert "replicated on every GPU (no data-path collective)" in par and "exchange" not in d
        return
    assert ("sharded by k-mer range over 2 GPUs" if db_mode == "shard" else "sharded by GENOME over 2 GPUs inside the library") in par
    legs = [d] + [d[k] for k in ("pipelined", "one_step_at_a_time") if k in d and "exchange" in d[k]]
    assert any("exchange" in leg and leg["exchange"]["probe_batches"] > 0 and leg["exchange"]["hit_bytes_sent_per_batch"] > 0 for leg in legs)


def test_bench_line_contract_one_gpu_small_workload():
    """`python bench.py` at N = 1 on a small workload: ONE JSON line with the driver's fields, the `roofline` object of the dominant kernel
    (bound, achieved, peak, frac, traffic key; since round 6 a pipeline's sample has two launches of the seeding kernel — head and tail — which
    the line says and prices), the `cpu_baseline` object (kind "port": the oracle, timed on the host), the side-by-side `rates_gbp_per_s`,
    a verify leg without mismatches, and sylph's default pair dedup beside the exact set."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "small", "--steps", "3", "--warmup", "1", "--min-seconds", "0.3",
                        "--cpu-baseline-bounded", "--no-files-leg", "--no-packed-leg"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["value"] > 0
    assert d["config"]["workload"] and d["data"].startswith("synthetic")
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and "traffic" in roof
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["avg_launch_ms"] > 0 and roof["launches"] > 0
    if roof.get("launches_per_sample", 1) > 1:                                   # the tail launch of a pipeline's turn (reads_tail_pct)
        ov = roof["overlapping_launches"]
        assert ov["sum_of_a_samples_launch_ms"] == roof["avg_launch_ms"] and ov["frac_per_wall_time"] > 0
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    assert d["verify"]["mismatches"] == 0 and d["verify"]["genomes_checked"] > 0
    assert d["rates_gbp_per_s"]["gpu_inputs_resident_exact_dedup"] == d["value"]
    if "default_pair_dedup" in d:
        assert d["value_default_flags"] > 0 and d["default_pair_dedup"]["verify"]["table_equal"] is True


def test_bench_two_ranks_genome_sharded_arm_composed_outside_the_library():
    """`bench.py --gpus 2 --db-mode genome-py`: round 4's composition of the cut by genome (sylph_amd/shard.py: unsharded index per rank,
    torch.distributed all-gathers of tables, counts and coverage values) — kept beside the library mode for the A/B."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT="29580")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "small", "--db-mode", "genome-py", "--steps", "2",
                        "--warmup", "1", "--min-seconds", "0.2", "--no-cpu-baseline", "--no-h2d"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["mode"] == "sequential"
    assert d["verify"]["mismatches"] == 0 and d["verify"]["genomes_checked"] > 0 and d["verify"]["genomes_with_hits"] > 0
    assert "sharded by GENOME" in d["config"]["parallelism"]


def test_one_sample_loop_over_several_replicas_of_the_database(ctx):
    """Round 5: sylph_db_replicate + sylph_pipeline_create_multi — ONE submit / next queue over N replicas of the database (one per GPU
    of a node; here, on the one-GPU box, two contexts of device 0).  Device batches and sessions go to a replica on their device, host
    batches to the least busy one; the results come back in SUBMISSION order whichever replica had them, each equal to the oracle's;
    the replicas answer a direct probe like the original (kept + tracked index copied device to device)."""
    import torch
    rng, genomes, db_k, goff = small_world(23)
    db = S.Database(ctx, db_k, goff)
    ctx2, ctx3 = S.Context(0), S.Context(0)
    # (the third replica by the long road: "fail_next_peer_copy" makes the copy into ctx3 find no device-to-device way — what two GPUs
    #  without peer access would do — and the index travels through the library's page-locked chunk; round 6)
    ctx3.set_option("fail_next_peer_copy", "1")
    reps = [db, db.replicate(ctx2), db.replicate(ctx3)]
    assert all(r.n_genomes == db.n_genomes and r.n_kmers == db.n_kmers for r in reps)
    samples = []
    for i in range(13):
        b, off = sample_reads(rng, genomes, [i % 5, (i * 3 + 1) % 5][: 1 + i % 2], 300 + 120 * (i % 4))
        samples.append((b, off, O.sketch_reads(b, off, c=50, paired=True)))
    # a replica probed directly gives what the original gives
    e0 = samples[0][2]
    for r in reps[1:]:
        cc, coff, covs = r.contain(e0["kmers"], e0["counts"])
        cc0, coff0, covs0 = db.contain(e0["kmers"], e0["counts"])
        assert np.array_equal(cc, cc0) and np.array_equal(coff, coff0) and np.array_equal(covs, covs0)
    dev = [(torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()) for b, off, _ in samples]
    torch.cuda.synchronize()
    p = S.Pipeline(reps, c=50, paired=True, n_workers=2, depth=3, max_batch=4, want_table=True)
    used = set()
    submitted = done = 0
    while done < len(samples):
        while submitted < len(samples) and p.outstanding < 3 * len(reps):
            b, off, _ = samples[submitted]
            if submitted % 3 == 0:          # host memory: the least busy replica
                ok = p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=submitted, mem=MEM_HOST)
            elif submitted % 3 == 1:        # device memory: a replica on that device
                ok = p.submit_device([(dev[submitted][0].data_ptr(), dev[submitted][1].data_ptr(), len(off) - 1, int(off[-1]))], tag=submitted)
            else:                           # a session the caller pushed into (on a context of its own, device 0)
                sk = S.ReadSketcher([ctx, ctx2, ctx3][submitted % 2], c=50, paired=True)
                sk.push(b, off)
                ok = p.submit_session(sk, tag=submitted)
            assert ok
            submitted += 1
        r = p.next()
        assert r["tag"] == done
        used.add(r["replica"])
        e = samples[done][2]
        check_result(r, e, db_k, goff)
        assert np.array_equal(r["kmers"], e["kmers"]) and np.array_equal(r["counts"], e["counts"])
        done += 1
    assert len(used) >= 2 and used <= {0, 1, 2}
    with pytest.raises(S.SylphHipError):
        p.next()
    # every replica full: refused, not blocked
    for i in range(3 * len(reps)):
        b, off, _ = samples[i]
        assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=i, mem=MEM_HOST)
    b, off, _ = samples[0]
    assert p.submit_device([(b.ctypes.data, off.ctypes.data, len(off) - 1, int(off[-1]))], tag=99, mem=MEM_HOST) is False
    p.close()                               # (finishes what is outstanding)
    for r in reps[1:]:
        r.close()
    db.close(); ctx2.close(); ctx3.close()
