"""One database over several GPUs: k-mer-range shards, one process per GPU.

The exchange itself lives in the library (csrc/shard.hip, `sylph_db_contain_batch_sharded`): slice boundaries all-gathered,
table slices all-to-all, one probe launch, hit groups all-to-all to the ranks that own the samples, local assembly — on device buffers, with
the collectives issued through a `sylph_comm` (RCCL, or callbacks).  This module holds what sits around it in Python:

* `rccl_comm(dist, ctx)`         — a Comm on RCCL: rank 0 makes the ncclUniqueId, torch.distributed (the launcher's rendezvous)
                                   hands it to every rank, the library creates its own communicator on its own stream;
* `torch_callback_comm(dist, …)` — a Comm whose collectives run through torch.distributed on ANY backend (gloo in the tests:
                                   device buffer -> host -> gloo -> device), so the library's exchange code runs without RCCL;
* `model_contain_batch_sharded`  — the same five steps written with numpy on host arrays and a pluggable probe: the executable
                                   specification of the protocol.  tests/test_dist.py runs it under gloo with world_size 2
                                   and 3 (the CPU oracle standing in for the HIP probe) against the single-process answer.
"""
import numpy as np
import torch


def shard_bounds(max_kmer, world):
    """Equal-width k-mer ranges over [0, max_kmer] (same arithmetic as sylph_shard_bounds): world + 1 boundaries."""
    span = int(max_kmer) + 1
    b = [span * r // world for r in range(world)] + [min(int(max_kmer) + 1, 2**64 - 1)]
    return np.array(b, dtype=np.uint64)


class LocalGroup:
    world, rank = 1, 0


class _DevArray:
    """Zero-copy torch view of library-owned device memory (cuda array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, n, dtype, device):
    if n == 0:
        return torch.zeros(0, dtype=dtype, device=device)
    typestr = {torch.int64: "<i8", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def rccl_comm(dist, ctx, device):
    """Comm over RCCL for the default process group's ranks.  The 128-byte id travels through torch.distributed."""
    from .binding import Comm
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == 0:
        t = torch.frombuffer(bytearray(Comm.rccl_unique_id()), dtype=torch.uint8).clone()
    else:
        t = torch.zeros(128, dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.to(device)
    dist.broadcast(t, src=0)
    return Comm(rank, world, ctx=ctx, rccl_id=bytes(t.cpu().numpy().tobytes()))


def torch_callback_comm(dist, device):
    """Comm whose collectives are torch.distributed calls on host copies of the device buffers (any backend)."""
    from .binding import Comm
    rank, world = dist.get_rank(), dist.get_world_size()

    def all_gather(send, recv, nbytes, stream):
        torch.cuda.synchronize()
        src = device_view(send, nbytes, torch.uint8, device).cpu()
        out = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(out, src)
        device_view(recv, world * nbytes, torch.uint8, device).copy_(torch.cat(out))
        torch.cuda.synchronize()

    def all_to_all(send, send_off, recv, recv_off, stream):
        torch.cuda.synchronize()
        total = send_off[world]
        src = device_view(send, total, torch.uint8, device).cpu().numpy() if total else np.zeros(0, np.uint8)
        blocks = [bytes(src[send_off[r]:send_off[r + 1]]) for r in range(world)]
        got = [None] * world
        dist.all_gather_object(got, blocks)
        mine = b"".join(got[r][rank] for r in range(world))
        assert len(mine) == recv_off[world], (len(mine), recv_off[world])
        if mine:
            device_view(recv, len(mine), torch.uint8, device).copy_(torch.frombuffer(bytearray(mine), dtype=torch.uint8))
        torch.cuda.synchronize()

    return Comm(rank, world, all_gather=all_gather, all_to_all=all_to_all)


def torch_device_comm(dist, device):
    """Comm whose collectives are torch.distributed calls ON the device buffers (backend nccl = RCCL): torch's own communicator
    instead of one created by the library.  The collectives are ordered with the library's stream by making it torch's current
    stream for the call."""
    from .binding import Comm
    rank, world = dist.get_rank(), dist.get_world_size()

    def all_gather(send, recv, nbytes, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0, device=device)):
            dist.all_gather_into_tensor(device_view(recv, world * nbytes, torch.uint8, device), device_view(send, nbytes, torch.uint8, device))

    def all_to_all(send, send_off, recv, recv_off, stream):
        ins = [int(send_off[r + 1] - send_off[r]) for r in range(world)]
        outs = [int(recv_off[r + 1] - recv_off[r]) for r in range(world)]
        with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0, device=device)):
            dist.all_to_all_single(device_view(recv, sum(outs), torch.uint8, device), device_view(send, sum(ins), torch.uint8, device), outs, ins)

    return Comm(rank, world, all_gather=all_gather, all_to_all=all_to_all)


def model_contain_batch_sharded(dist, bounds, n_genomes, samples, probe_fn):
    """Host model of sylph_db_contain_batch_sharded (see csrc/shard.hip), step for step.

    dist: torch.distributed (initialised) or None for a single process; bounds: the k-mer range boundaries (world + 1);
    samples: this rank's [(kmers uint64 ascending, counts uint32)]; probe_fn(kmers, counts) -> [(genome, count)] hits of one
    slice against this rank's shard (every posting of a k-mer yields one hit).
    -> (contain_count[n_local, n_genomes] uint32, covs: list (per sample) of lists (per genome) of ascending uint32 arrays)."""
    world = dist.get_world_size() if dist is not None else 1
    me = dist.get_rank() if dist is not None else 0

    def all_gather_obj(x):
        if dist is None:
            return [x]
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    bounds = np.asarray(bounds, dtype=np.uint64)
    # 1. slice boundaries of every local table, all-gathered
    split = [np.searchsorted(k, bounds, side="left") for k, _ in samples]
    meta = all_gather_obj(split)
    prefix = np.concatenate([[0], np.cumsum([len(m) for m in meta])])
    # 2. all-to-all of the slices (modelled as an all-gather of the per-destination blocks)
    blocks = [[(k[sp[d]:sp[d + 1]], c[sp[d]:sp[d + 1]]) for (k, c), sp in zip(samples, split)] for d in range(world)]
    got = all_gather_obj(blocks)
    # 3. probe every received slice; row = global sample index * G + genome
    hits = []
    for r in range(world):
        for s, (k, c) in enumerate(got[r][me]):
            assert len(k) == meta[r][s][me + 1] - meta[r][s][me]
            row0 = (int(prefix[r]) + s) * n_genomes
            hits += [((row0 + g) << 32) | int(cnt) for g, cnt in probe_fn(k, c)]
    # 4. group the hits by the rank that owns their sample; all-gather the group sizes
    def owner(h):
        s = (h >> 32) // n_genomes
        return int(np.searchsorted(prefix, s, side="right")) - 1
    groups = [[] for _ in range(world)]
    for h in hits:
        r = owner(h)
        groups[r].append(h - ((int(prefix[r]) * n_genomes) << 32))        # rows re-based to the owner's samples
    sizes = all_gather_obj([len(g) for g in groups])
    assert all(sizes[me][r] == len(groups[r]) for r in range(world))
    # 5. all-to-all of the hit groups (modelled as an all-gather of the per-destination groups): every rank keeps what is addressed to it
    got_hits = all_gather_obj(groups)
    mine = sorted(h for r in range(world) for h in got_hits[r][me])
    assert len(mine) == sum(sizes[r][me] for r in range(world))
    # 6. sort + assemble
    n_local = len(samples)
    cc = np.zeros((n_local, n_genomes), dtype=np.uint32)
    covs = [[[] for _ in range(n_genomes)] for _ in range(n_local)]
    for h in mine:
        row, cnt = h >> 32, h & 0xFFFFFFFF
        cc[row // n_genomes, row % n_genomes] += 1
        covs[row // n_genomes][row % n_genomes].append(cnt)
    return cc, [[np.array(x, dtype=np.uint32) for x in per] for per in covs]


# ---- the other way to shard (north_star's wording): by GENOME, results reduced by one all-gather of containment counts -------------
#
# Every rank holds the whole sketches of a contiguous range of genomes (balanced by k-mer count) as an ordinary, unsharded database
# of its own; every sample table must then be probed by EVERY rank (per-rank probe work per sample does not shrink with the number
# of GPUs — why the library's default is the k-mer-range exchange of csrc/shard.hip), and what comes back is, per rank, the
# containment counts of its genomes for all samples of the step: one all-gather of contain_count[S, G/W] (+ an all-gather of the
# coverage values, sizes first) gives every rank the full answer for its own samples.  This arm exists for the A/B on the first
# 8-GPU node (bench.py --db-mode genome); it is composed from the unsharded entry points + torch.distributed collectives (RCCL on
# device tensors), not a second exchange inside the library.

def genome_shard_bounds(genome_off, world):
    """sylph_genome_shard_bounds: the library's own cut (csrc/shard_plan.h) -> world + 1 genome indices (uint64)."""
    import ctypes as C
    from .binding import _check, _np, _ptr, load
    off = _np(genome_off, np.uint64)
    out = np.zeros(world + 1, dtype=np.uint64)
    _check(load().sylph_genome_shard_bounds(_ptr(off), len(off) - 1, world, _ptr(out)))
    return out


def genome_shard_ranges(genome_off, world):
    """Contiguous genome ranges [g0, g1) per rank, balanced by the number of k-mers: world + 1 boundaries."""
    off = np.asarray(genome_off, dtype=np.uint64)
    G, total = len(off) - 1, int(off[-1])
    b = [0]
    for r in range(1, world):
        b.append(int(np.searchsorted(off, total * r // world, side="left")))
    b.append(G)
    for i in range(1, len(b)):
        b[i] = max(b[i], b[i - 1])
    return np.array(b, dtype=np.int64)


def contain_batch_genome_sharded(dist, db_local, ranges, refs, device, min_number_kmers=50.0):
    """One step of the genome-sharded arm.  refs: this rank's [(dev_kmers_ptr, dev_counts_ptr, n)] (device-resident tables, the same
    NUMBER on every rank); db_local: an unsharded Database over genomes [ranges[rank], ranges[rank + 1]).
    -> (contain_count[n_local * G] uint32, cov_off[n_local * G + 1] int64, covs uint32) for this rank's samples, genomes in global order
    (row = sample * G + genome: the layout of Database.contain_batch)."""
    W, me = dist.get_world_size(), dist.get_rank()
    n_local = len(refs)
    G = int(ranges[-1])
    on_dev = dist.get_backend() == "nccl"
    cdev = device if on_dev else torch.device("cpu")

    def gather(t):                                  # fixed-size all-gather of a tensor -> [W, ...]
        t = t.to(cdev).contiguous()
        out = torch.empty((W,) + tuple(t.shape), dtype=t.dtype, device=cdev)
        dist.all_gather_into_tensor(out, t) if on_dev else dist.all_gather(list(out.unbind(0)), t)
        return out

    # 1. table sizes, then the tables themselves (padded to the longest one: all-gather wants equal blocks)
    sizes = gather(torch.tensor([r[2] for r in refs], dtype=torch.int64))                    # [W, n_local]
    n_max = max(1, int(sizes.max().item()))
    k_blk = torch.zeros((n_local, n_max), dtype=torch.int64, device=device)
    c_blk = torch.zeros((n_local, n_max), dtype=torch.int32, device=device)
    for s, (kp, cp, n) in enumerate(refs):
        if n:
            k_blk[s, :n] = device_view(kp, n, torch.int64, device)
            c_blk[s, :n] = device_view(cp, n, torch.int32, device)
    all_k, all_c = gather(k_blk).to(device), gather(c_blk).to(device)                        # [W, n_local, n_max]
    # 2. every table of the step against this rank's genomes
    S_total = W * n_local
    tabs = [(all_k[r, s].data_ptr(), all_c[r, s].data_ptr(), int(sizes[r, s].item())) for r in range(W) for s in range(n_local)]
    torch.cuda.synchronize(device) if device.type == "cuda" else None
    cc, off, covs = db_local.contain_batch(tabs, min_number_kmers=min_number_kmers, device_ptrs=True)
    G_loc = int(ranges[me + 1] - ranges[me])
    cc = torch.from_numpy(np.array(cc, dtype=np.uint32).astype(np.int32)).view(S_total, G_loc)
    covs = torch.from_numpy(np.asarray(covs).astype(np.int32))
    off = np.array(off, dtype=np.int64)
    # 3. THE all-gather of north_star: contain_count[S, G/W] of every shard (padded to the widest shard)
    G_max = int(max(ranges[r + 1] - ranges[r] for r in range(W)))
    cc_pad = torch.zeros((S_total, G_max), dtype=torch.int32)
    cc_pad[:, :G_loc] = cc
    all_cc = gather(cc_pad).cpu().numpy().astype(np.uint32)                                  # [W, S_total, G_max]
    # 4. the coverage values: sizes first, then the payload padded to the longest list
    n_cov = gather(torch.tensor([covs.numel()], dtype=torch.int64)).cpu().numpy().reshape(-1)
    cov_pad = torch.zeros(max(1, int(n_cov.max())), dtype=torch.int32)
    cov_pad[:covs.numel()] = covs
    all_cov = gather(cov_pad).cpu().numpy().astype(np.uint32)                                # [W, max]
    # 5. this rank's samples, genomes back in global order (rows of one sample are contiguous in every shard's block)
    out_cc = np.zeros((n_local, G), dtype=np.uint32)
    pieces = []
    for s in range(n_local):
        row = me * n_local + s
        for r in range(W):
            g0, g1 = int(ranges[r]), int(ranges[r + 1])
            counts = all_cc[r, row, :g1 - g0]
            out_cc[s, g0:g1] = counts
            start = int(all_cc[r, :row, :g1 - g0].sum())                                    # hits of the shard's rows before this sample
            pieces.append(all_cov[r, start:start + int(counts.sum())])
    out_cov = np.concatenate(pieces) if pieces else np.zeros(0, np.uint32)
    out_off = np.zeros(n_local * G + 1, dtype=np.int64)
    out_off[1:] = np.cumsum(out_cc.reshape(-1).astype(np.int64))
    return out_cc.reshape(-1), out_off, out_cov


def model_contain_batch_genome_sharded(dist, ranges, samples, probe_fn):
    """The same protocol on host arrays with a pluggable probe (tests/test_dist.py, gloo): samples = this rank's [(kmers, counts)],
    probe_fn(kmers, counts) -> (contain_count[G_local], [sorted covs per local genome]).  -> (contain_count[n_local, G], covs[s][g])."""
    W, me = dist.get_world_size(), dist.get_rank()
    got = [None] * W
    dist.all_gather_object(got, samples)                                  # every table to every rank
    n_local = len(samples)
    mine = [[probe_fn(k, c) for (k, c) in got[r]] for r in range(W)]      # [owner rank][sample] -> (cc_local, covs_local)
    allres = [None] * W
    dist.all_gather_object(allres, mine)                                  # "one all-gather of the containment counts" (+ covs)
    G = int(ranges[-1])
    cc = np.zeros((n_local, G), dtype=np.uint32)
    covs = [[None] * G for _ in range(n_local)]
    for s in range(n_local):
        for r in range(W):
            ccl, covl = allres[r][me][s]
            g0 = int(ranges[r])
            cc[s, g0:g0 + len(ccl)] = ccl
            for j, v in enumerate(covl):
                covs[s][g0 + j] = np.asarray(v, dtype=np.uint32)
    return cc, covs
