"""N>1 path on CPU: world_size-2 and -3 gloo runs of the k-mer-range sharded containment exchange.  The product's exchange runs
in the HIP library (csrc/shard.hip) and needs a GPU; what runs here is its executable specification,
sylph_amd.shard.model_contain_batch_sharded — the same five steps on host arrays — with the oracle standing in for the HIP probe.
Every rank must recover, for its OWN samples, exactly the single-process answer over the WHOLE database.  (tests/test_gpu_parity.py
runs the library's exchange itself over gloo with two ranks on one GPU, and over RCCL with one rank.)"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import sylph_amd as S
from oracle import oracle as O
from sylph_amd import shard as SH


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_db(seed=5, G=37):
    rng = np.random.default_rng(seed)
    thr = O.threshold(200)
    pool = np.unique(rng.integers(0, thr, size=30000, dtype=np.uint64))
    lens = rng.integers(0, 900, size=G)
    lens[3] = 0
    lens[7] = 49
    genomes = [rng.choice(pool, size=int(n), replace=False) for n in lens]
    genomes[11] = np.concatenate([genomes[11], genomes[11][:5]])       # a k-mer twice in one genome: counted twice (contain.rs:632)
    return pool, genomes


def make_samples(pool, rank, n):
    rng = np.random.default_rng(100 + rank)
    out = []
    for i in range(n):
        k = np.sort(rng.choice(pool, size=3000 + 500 * rank + 100 * i, replace=False)) if (rank, i) != (1, 1) else np.zeros(0, np.uint64)
        out.append((k, rng.integers(0, 9, size=len(k)).astype(np.uint32)))
    return out


def shard_probe(genomes, lo, hi, min_number_kmers=50.0):
    """CPU stand-in for the resident shard: genomes cut to the k-mer range, full genome lengths kept for the :627 test."""
    cut = [g[(g >= lo) & ((g < hi) if hi else True)] for g in genomes]
    off = np.zeros(len(cut) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(g) for g in cut])
    flat = np.concatenate(cut) if len(cut) else np.zeros(0, np.uint64)
    full_len = [len(g) for g in genomes]

    def probe(k, c):
        cc, covs, _ = O.contain(k, c, flat, off, min_number_kmers=0.0)
        return [(g, int(x)) for g in range(len(cut)) if full_len[g] >= min_number_kmers for x in covs[g]]
    return probe


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool, genomes = make_db()
        G = len(genomes)
        bounds = SH.shard_bounds(int(max(g.max() for g in genomes if len(g))), world)
        hi = int(bounds[rank + 1]) if rank + 1 < world else 0
        samples = make_samples(pool, rank, [2, 3, 0][rank])      # different batch sizes per rank, one rank may bring none
        cc, covs = SH.model_contain_batch_sharded(dist, bounds, G, samples, shard_probe(genomes, int(bounds[rank]), hi))
        db = np.concatenate(genomes)
        goff = np.zeros(G + 1, dtype=np.uint64)
        goff[1:] = np.cumsum([len(g) for g in genomes])
        ok = cc.shape == (len(samples), G)
        for s, (k, c) in enumerate(samples):
            ecc, ecov, _ = O.contain(k, c, db, goff)
            ok = ok and np.array_equal(cc[s], ecc) and all(np.array_equal(covs[s][g], np.sort(ecov[g])) for g in range(G))
            ok = ok and (len(k) == 0 or int(ecc.sum()) > 0)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_containment_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world


def test_shard_bounds_match_library_and_cover_the_range():
    for mx in (0, 1, 6, 2**64 // 200, 2**64 - 2, 2**64 - 1):
        for w in (1, 2, 3, 8):
            b = SH.shard_bounds(mx, w)
            assert np.array_equal(b, S.shard_bounds(mx, w)), (mx, w)          # host function of the C ABI: runs without a GPU
            assert b[0] == 0 and all(b[i] <= b[i + 1] for i in range(w)) and int(b[w]) >= min(mx, 2**64 - 2)
    b = [int(x) for x in SH.shard_bounds(2**64 // 200, 8)]
    w = [b[i + 1] - b[i] for i in range(8)]
    assert max(w) - min(w) <= 2          # equal widths: uniform hashes => equal postings and equal slices per rank


# ---- the library's own exchange bookkeeping (csrc/shard_plan.h, what shard.hip runs between its collectives) under gloo ------------

def _plan_lib():
    """tests/shard_plan_capi.cpp + the header shard.hip includes, compiled with g++ (no HIP) into a scratch directory."""
    import ctypes as C
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(tempfile.gettempdir(), f"sylph_shard_plan_{os.getuid()}.so")
    src = os.path.join(here, "shard_plan_capi.cpp")
    hdr = os.path.join(here, "..", "sylph_amd", "csrc", "shard_plan.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        tmp = out + f".{os.getpid()}"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", tmp, src])
        os.replace(tmp, out)
    L = C.CDLL(out)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    L.sp_meta_words.restype = C.c_uint64
    L.sp_lower_bound.restype = C.c_uint64
    L.sp_lower_bound.argtypes = [u64p, C.c_uint64, C.c_uint64]
    L.sp_owner.argtypes = [C.c_uint64, u64p, C.c_uint32]
    L.sp_rebase.restype = C.c_uint64
    L.sp_rebase.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.sp_plan_slices.argtypes = [u64p, C.c_uint32, C.c_uint32, C.c_uint64, u64p, u64p, u64p, u64p, C.c_char_p, C.c_size_t]
    L.sp_slice_in_block.argtypes = [u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p]
    L.sp_plan_hits.argtypes = [u32p, C.c_uint32, C.c_uint32, u64p, u64p, u32p, u32p, u64p, u32p, u32p, C.c_char_p, C.c_size_t]
    L.sp_set_whole.argtypes = [C.c_int]
    L.sp_slice_begin.restype = C.c_uint64
    L.sp_slice_begin.argtypes = [u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.sp_genome_bounds.argtypes = [u64p, C.c_uint64, C.c_uint32, u64p]
    L.sp_plan_hits_gather.argtypes = [u32p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), u64p, u64p]
    L.sp_plan_hits_gather.restype = None
    return L


def _p64(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _p32(a):
    import ctypes as C
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def library_plan_exchange(dist, L, bounds, G, samples, probe_fn, fail=False, reduce="alltoall"):
    """sylph_db_contain_batch_sharded's steps on host arrays, EVERY offset, owner and size taken from csrc/shard_plan.h (through
    tests/shard_plan_capi.cpp) exactly as shard.hip takes them; byte buffers laid out as the device buffers are; the collectives are
    gloo's.  -> (contain_count[n_local, G], covs) or the (rank, class) of the agreed failure."""
    import ctypes as C
    W, me = dist.get_world_size(), dist.get_rank()
    ML, mw = L.sp_max_local(), int(L.sp_meta_words(W))
    bounds = np.ascontiguousarray(bounds, dtype=np.uint64)
    # 1. my meta block [n_local | split[MAX_LOCAL][W + 1]], all-gathered
    block = np.zeros(mw, dtype=np.uint64)
    block[0] = len(samples)
    for s, (k, _) in enumerate(samples):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        for j in range(W + 1):
            block[1 + s * (W + 1) + j] = L.sp_lower_bound(_p64(k), len(k), int(bounds[j]))
    import torch
    got = [torch.zeros(mw, dtype=torch.int64) for _ in range(W)]
    dist.all_gather(got, torch.from_numpy(block.view(np.int64)))
    meta = np.ascontiguousarray(np.concatenate([g.numpy().view(np.uint64) for g in got]))
    prefix, send_off, recv_off = (np.zeros(W + 1, dtype=np.uint64) for _ in range(3))
    S_total = C.c_uint64(0)
    err = C.create_string_buffer(256)
    assert L.sp_plan_slices(_p64(meta), W, me, G, _p64(prefix), _p64(send_off), _p64(recv_off), C.byref(S_total), err, 256) == 0, err.value
    # 2. pack my send buffer block by block, slice by slice, at the offsets the plan gives; all-to-all
    send = np.zeros(int(send_off[W]), dtype=np.uint8)
    at = np.zeros(3, dtype=np.uint64)
    for d in range(W):
        for s, (k, c) in enumerate(samples):
            L.sp_slice_in_block(_p64(meta), W, me, s, d, _p64(at))
            ko, co, ln = int(at[0]), int(at[1]), int(at[2])
            a = int(L.sp_slice_begin(_p64(meta), W, me, s, d))      # (k-mer ranges: split[s][d]; genome shards: 0 — the whole table)
            base = int(send_off[d])
            send[base + ko:base + ko + 8 * ln] = np.ascontiguousarray(k[a:a + ln], dtype=np.uint64).view(np.uint8)
            send[base + co:base + co + 4 * ln] = np.ascontiguousarray(c[a:a + ln], dtype=np.uint32).view(np.uint8)
    blocks = [bytes(send[int(send_off[r]):int(send_off[r + 1])]) for r in range(W)]
    allb = [None] * W
    dist.all_gather_object(allb, blocks)
    recv = np.frombuffer(b"".join(allb[r][me] for r in range(W)), dtype=np.uint8)
    assert len(recv) == int(recv_off[W])
    # 3. probe every received slice where the plan says it lies; row = global sample * G + genome
    hits = []
    n_loc = [int(meta[r * mw]) for r in range(W)]
    for r in range(W):
        for s in range(n_loc[r]):
            L.sp_slice_in_block(_p64(meta), W, r, s, me, _p64(at))
            ko, co, ln = int(at[0]), int(at[1]), int(at[2])
            base = int(recv_off[r])
            k = recv[base + ko:base + ko + 8 * ln].view(np.uint64)
            c = recv[base + co:base + co + 4 * ln].view(np.uint32)
            row0 = (int(prefix[r]) + s) * G
            hits += [((row0 + g) << 32) | int(cnt) for g, cnt in probe_fn(k, c)]
    hits = np.array(hits, dtype=np.uint64)
    # 4. owner counts + trailer, all-gathered; the plan's verdict
    SZ = W + 3
    sizes = np.zeros(SZ, dtype=np.uint32)
    owners = np.array([L.sp_owner(int(h >> np.uint64(32)) // G, _p64(prefix), W) for h in hits], dtype=np.int64)
    if fail:
        hits, owners = hits[:0], owners[:0]
    for r in range(W):
        sizes[r] = int((owners == r).sum())
    sizes[W] = int((hits & np.uint64(0xFFFFFFFF)).max()) if len(hits) else 0
    sizes[W + 1] = 1 if fail else 0
    sizes[W + 2] = len(hits)
    gots = [torch.zeros(SZ, dtype=torch.int32) for _ in range(W)]
    dist.all_gather(gots, torch.from_numpy(sizes.view(np.int32)))
    allsizes = np.ascontiguousarray(np.concatenate([g.numpy().view(np.uint32) for g in gots]))
    hs_off, hr_off = np.zeros(W + 1, dtype=np.uint64), np.zeros(W + 1, dtype=np.uint64)
    start = np.zeros(W, dtype=np.uint32)
    max_mine, n_mine, fr, fc = C.c_uint32(0), C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
    rc = L.sp_plan_hits(_p32(allsizes), W, me, _p64(hs_off), _p64(hr_off), _p32(start), C.byref(max_mine), C.byref(n_mine), C.byref(fr), C.byref(fc), err, 256)
    if rc == 2:
        return ("failed", int(fr.value), int(fc.value))
    assert rc == 0, err.value
    # 5. scatter the hits into the send buffer at start[r] (+ running cursor), rows re-based; all-to-all
    out = np.zeros(int(hs_off[W]) // 8, dtype=np.uint64)
    cur = start.astype(np.int64).copy()
    for h, r in zip(hits, owners):
        out[cur[r]] = L.sp_rebase(int(h), int(prefix[r]), G)
        cur[r] += 1
    if reduce == "alltoall":
        groups = [out[int(hs_off[r]) // 8:int(hs_off[r + 1]) // 8].tobytes() for r in range(W)]
        allg = [None] * W
        dist.all_gather_object(allg, groups)
        mine = np.frombuffer(b"".join(allg[r][me] for r in range(W)), dtype=np.uint64)
    else:
        # "shard_reduce" = "allgather": every rank's WHOLE grouped buffer, padded to the longest, to every rank in ONE all-gather of equal
        # blocks (a real tensor collective here, as RCCL's would be); my group of each block where shard_plan.h's plan_hits_gather says
        pad, src_off, ln = C.c_uint64(0), np.zeros(W, dtype=np.uint64), np.zeros(W, dtype=np.uint64)
        L.sp_plan_hits_gather(_p32(allsizes), W, me, C.byref(pad), _p64(src_off), _p64(ln))
        block = np.full(int(pad.value), 0xEE, dtype=np.uint8)                 # (the padding is never read)
        block[:len(out) * 8] = out.view(np.uint8)
        got = [torch.zeros(int(pad.value), dtype=torch.uint8) for _ in range(W)]
        if pad.value:
            dist.all_gather(got, torch.from_numpy(block))
        gathered = np.concatenate([g.numpy() for g in got]) if pad.value else np.zeros(0, dtype=np.uint8)
        mine_b = np.zeros(int(hr_off[W]), dtype=np.uint8)
        for r in range(W):
            mine_b[int(hr_off[r]):int(hr_off[r]) + int(ln[r])] = gathered[int(src_off[r]):int(src_off[r]) + int(ln[r])]
            assert int(ln[r]) == int(hr_off[r + 1]) - int(hr_off[r])
        mine = mine_b.view(np.uint64)
    assert len(mine) == int(n_mine.value) == int(hr_off[W]) // 8
    assert (int((mine & np.uint64(0xFFFFFFFF)).max()) if len(mine) else 0) <= int(max_mine.value)
    # 6. assemble
    n_local = len(samples)
    cc = np.zeros((n_local, G), dtype=np.uint32)
    covs = [[[] for _ in range(G)] for _ in range(n_local)]
    for h in np.sort(mine):
        row, cnt = int(h) >> 32, int(h) & 0xFFFFFFFF
        cc[row // G, row % G] += 1
        covs[row // G][row % G].append(cnt)
    return cc, [[np.array(x, dtype=np.uint32) for x in per] for per in covs]


def _plan_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = _plan_lib()
        pool, genomes = make_db()
        G = len(genomes)
        bounds = SH.shard_bounds(int(max(g.max() for g in genomes if len(g))), world)
        hi = int(bounds[rank + 1]) if rank + 1 < world else 0
        probe = shard_probe(genomes, int(bounds[rank]), hi)
        db = np.concatenate(genomes)
        goff = np.zeros(G + 1, dtype=np.uint64)
        goff[1:] = np.cumsum([len(g) for g in genomes])
        ok = True
        for step, sizes in enumerate(([2, 3, 0], [0, 1, 4], [1, 1, 1])):     # three batches of different shapes, an empty rank, an empty table
            samples = make_samples(pool, rank + 3 * step, sizes[rank]) if step else make_samples(pool, rank, sizes[rank])
            for reduce in ("alltoall", "allgather"):          # (round 6: the hits by all-to-all, and by ONE all-gather of padded blocks)
                cc, covs = library_plan_exchange(dist, L, bounds, G, samples, probe, reduce=reduce)
                ok = ok and cc.shape == (len(samples), G)
                for s, (k, c) in enumerate(samples):
                    ecc, ecov, _ = O.contain(k, c, db, goff)
                    ok = ok and np.array_equal(cc[s], ecc) and all(np.array_equal(covs[s][g], np.sort(ecov[g])) for g in range(G))
        # a rank that fails between the collectives: every rank must come out with the same (rank, class)
        res = library_plan_exchange(dist, L, bounds, G, make_samples(pool, rank, 1), probe, fail=(rank == world - 1))
        ok = ok and res == ("failed", world - 1, 1)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_library_exchange_bookkeeping_gloo(world):
    """csrc/shard_plan.h — the split / offset / owner / size arithmetic shard.hip runs between its collectives — compiled for the
    host and driven rank against rank over gloo: three batches of different shapes per world size, buffers laid out byte for byte as
    the device buffers, every rank's own samples against the single-process oracle over the whole database; and the agreed
    failure (one rank raises its error word: all ranks return the same verdict)."""
    _plan_lib()                                   # compile once, before the ranks race for it
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_plan_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world


def genome_shard_probe(genomes, g0, g1, min_number_kmers=50.0):
    """CPU stand-in for a GENOME shard (sylph_db_upload_genome_shard): all k-mers of the genomes [g0, g1), global ids."""
    inner = shard_probe([g if g0 <= i < g1 else g[:0] for i, g in enumerate(genomes)], 0, 0, 0.0)
    full_len = [len(g) for g in genomes]
    return lambda k, c: [(g, x) for g, x in inner(k, c) if full_len[g] >= min_number_kmers]


def _genome_plan_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = _plan_lib()
        L.sp_set_whole(1)
        pool, genomes = make_db()
        G = len(genomes)
        db = np.concatenate(genomes)
        goff = np.zeros(G + 1, dtype=np.uint64)
        goff[1:] = np.cumsum([len(g) for g in genomes])
        gb = np.zeros(world + 1, dtype=np.uint64)
        L.sp_genome_bounds(_p64(goff), G, world, _p64(gb))
        ok = int(gb[0]) == 0 and int(gb[world]) == G and all(gb[i] <= gb[i + 1] for i in range(world))
        per = [int(goff[int(gb[i + 1])] - goff[int(gb[i])]) for i in range(world)]
        ok = ok and max(per) - min(per) <= 2 * 900                          # within a genome or two of each other
        probe = genome_shard_probe(genomes, int(gb[rank]), int(gb[rank + 1]))
        bounds = np.full(world + 1, np.iinfo(np.uint64).max, dtype=np.uint64)   # what the library sets for a genome shard: {0, ~0, ..}
        bounds[0] = 0
        for step, sizes in enumerate(([2, 3, 0], [0, 1, 4], [1, 1, 1])):
            samples = make_samples(pool, rank + 3 * step, sizes[rank]) if step else make_samples(pool, rank, sizes[rank])
            for reduce in ("alltoall", "allgather"):          # north_star's shape: whole genomes per rank AND one all-gather of the answers
                cc, covs = library_plan_exchange(dist, L, bounds, G, samples, probe, reduce=reduce)
                ok = ok and cc.shape == (len(samples), G)
                for s, (k, c) in enumerate(samples):
                    ecc, ecov, _ = O.contain(k, c, db, goff)
                    ok = ok and np.array_equal(cc[s], ecc) and all(np.array_equal(covs[s][g], np.sort(ecov[g])) for g in range(G))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_library_genome_shards_bookkeeping_gloo(world):
    """Round 5: the database cut by GENOME inside the library (sylph_db_upload_genome_shard; north_star's wording) — the same exchange
    with every table travelling whole to every shard (shard_plan.h Meta::whole) and the genome ranges of sylph_genome_shard_bounds: the
    header's arithmetic rank against rank over gloo, every rank's own samples against the single-process oracle."""
    _plan_lib()
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_genome_plan_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world


# ---- the genome-sharded arm (north_star's wording; bench.py --db-mode genome) -----------------------------------------------------

def _genome_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pool, genomes = make_db()
        G = len(genomes)
        goff = np.zeros(G + 1, dtype=np.uint64)
        goff[1:] = np.cumsum([len(g) for g in genomes])
        ranges = SH.genome_shard_ranges(goff, world)
        g0, g1 = int(ranges[rank]), int(ranges[rank + 1])
        loc = genomes[g0:g1]
        lflat = np.concatenate(loc) if loc and sum(len(g) for g in loc) else np.zeros(0, np.uint64)
        loff = np.zeros(len(loc) + 1, dtype=np.uint64)
        loff[1:] = np.cumsum([len(g) for g in loc])

        def probe(k, c):
            cc, covs, _ = O.contain(k, c, lflat, loff)
            return cc, [np.sort(x) for x in covs]
        samples = make_samples(pool, rank, 2)                 # (the same number on every rank: what the arm's fixed-size all-gathers need)
        cc, covs = SH.model_contain_batch_genome_sharded(dist, ranges, samples, probe)
        db = np.concatenate(genomes)
        ok = cc.shape == (2, G) and int(ranges[0]) == 0 and int(ranges[-1]) == G and all(ranges[i] <= ranges[i + 1] for i in range(world))
        for s, (k, c) in enumerate(samples):
            ecc, ecov, _ = O.contain(k, c, db, goff)
            ok = ok and np.array_equal(cc[s], ecc) and all(np.array_equal(covs[s][g], np.sort(ecov[g])) for g in range(G))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_genome_sharded_arm_gloo(world):
    """Genomes split into contiguous ranges balanced by k-mer count, every table probed by every rank, one all-gather of the per-shard
    containment counts (+ coverage values): each rank recovers the single-process answer for its own samples."""
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_genome_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == [True] * world


def test_genome_shard_ranges_balance():
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 5000, size=1000)
    off = np.zeros(1001, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    for w in (1, 2, 3, 8):
        r = SH.genome_shard_ranges(off, w)
        assert r[0] == 0 and r[-1] == 1000 and len(r) == w + 1 and all(r[i] <= r[i + 1] for i in range(w))
        per = [int(off[r[i + 1]] - off[r[i]]) for i in range(w)]
        assert max(per) - min(per) <= 2 * 5000            # within one genome of each other
