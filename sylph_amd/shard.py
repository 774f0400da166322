"""Multi-GPU containment: database sharded by genome, one process per GPU, RCCL over xGMI.

Per step every rank owns one sample.  The (small, <= ~60 MB) sample tables are exchanged with one all-gather, every
rank probes all samples of the step against its resident shard, and the per-shard containment counts + coverage
lists are combined with ONE all-gather of a fixed-layout result buffer (counts, then the packed coverage lists).
Payloads are a few MB, so this is latency-bound, not xGMI-link-bound; nothing here translates an NCCL call pattern.

The functions take a `contain_fn(kmers_tensor, counts_tensor) -> (contain_count, cov_off, covs)` so that the same
exchange code runs on RCCL with the HIP kernels and — in tests/test_dist.py — on gloo with a CPU stand-in.
"""
import numpy as np
import torch


def partition_genomes(lens, world):
    """owner[g] for every genome: genomes sorted by k-mer count (desc) are dealt to ranks in snake order, which
    balances both the number of genomes and the number of k-mers per shard.  Deterministic on every rank."""
    lens = np.asarray(lens)
    owner = np.zeros(len(lens), dtype=np.int32)
    if world <= 1 or len(lens) == 0:
        return owner
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    i = np.arange(len(lens))
    r = i % (2 * world)
    owner[order] = np.where(r < world, r, 2 * world - 1 - r).astype(np.int32)
    return owner


class LocalGroup:
    world, rank = 1, 0


class TorchGroup:
    """torch.distributed process group (backend "nccl" == RCCL on ROCm, or "gloo" on CPU)."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def all_gather_fixed(self, t):
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return out

    def all_gather_flat(self, t, out):
        """One collective into a preallocated [world * n] tensor (no per-step allocations, no stacking)."""
        try:
            self.dist.all_gather_into_tensor(out, t)
        except (RuntimeError, AttributeError, NotImplementedError):   # backend without the flat variant
            parts = self.all_gather_fixed(t)
            out.copy_(torch.cat(parts))
        return out

    def all_gather_var(self, t):
        """all-gather of 1-D tensors of different lengths: sizes first, then one padded payload."""
        n = torch.tensor([t.numel()], dtype=torch.int64, device=self.device)
        sizes = [int(x.item()) for x in self.all_gather_fixed(n)]
        m = max(max(sizes), 1)
        pad = torch.zeros(m, dtype=t.dtype, device=self.device)
        pad[: t.numel()] = t
        return [o[:s] for o, s in zip(self.all_gather_fixed(pad), sizes)]


class _DevArray:
    """Zero-copy torch view of library-owned device memory (cuda array interface)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, n, dtype, device):
    if n == 0:
        return torch.zeros(0, dtype=dtype, device=device)
    typestr = {torch.int64: "<i8", torch.int32: "<i4"}[dtype]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


def to_host_numpy(t):
    """Device tensor -> numpy through PINNED host memory (torch's caching host allocator).  A plain `.cpu()` lands in
    pageable memory that the HIP runtime registers; when numpy frees it the driver evicts this process's GPU queues
    (15-40 ms stalls) — see DESIGN.md §3."""
    if not t.is_cuda:
        return t.numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=False)
    return h.numpy()


def exchange_and_profile(contain_fn, group, sample_k, sample_c, owner, rank_genomes):
    """sample_k (int64 bit patterns of u64) / sample_c (int32 bit patterns of u32): this rank's sample table.
    owner[g]: rank that holds genome g.  rank_genomes[r]: global ids of rank r's genomes in shard order.
    -> dict(contain_count[n_total] uint32, cov_off[n_total+1] uint64, covs uint32) for THIS rank's sample."""
    n_total = len(owner)
    world, rank = group.world, group.rank
    if world == 1:
        cc, off, covs = contain_fn(sample_k, sample_c)
        return dict(contain_count=np.asarray(cc), cov_off=np.asarray(off), covs=np.asarray(covs))
    # 1. exchange sample tables (all ranks see every sample of the step)
    ks = group.all_gather_var(sample_k)
    cs = group.all_gather_var(sample_c)
    # 2. probe every sample against the resident shard
    G_local = len(rank_genomes[rank])
    G_max = max(len(g) for g in rank_genomes)
    parts, cov_parts = [], []
    for s in range(world):
        cc, off, covs = contain_fn(ks[s], cs[s])
        row = np.zeros(G_max, dtype=np.int64)
        row[:G_local] = cc
        parts.append(row)
        cov_parts.append(np.array(covs, dtype=np.int64))   # a copy: contain_fn may hand out views it will overwrite
    # 3. ONE all-gather of [counts of all samples | packed covs of all samples] per rank
    counts_block = np.concatenate(parts)
    payload = np.concatenate([counts_block] + cov_parts)
    dev = group.device
    got = group.all_gather_var(torch.from_numpy(payload).to(dev))
    # 4. assemble this rank's sample in global genome order
    contain_count = np.zeros(n_total, dtype=np.uint32)
    cov_lists = [None] * world
    for r in range(world):
        buf = to_host_numpy(got[r])
        counts = buf[: world * G_max].reshape(world, G_max)
        covs_r = buf[world * G_max:]
        g_r = len(rank_genomes[r])
        # covs of sample s from rank r start after the covs of samples < s
        start = int(counts[:rank, :g_r].sum())
        mine = counts[rank, :g_r]
        contain_count[rank_genomes[r]] = mine.astype(np.uint32)
        cov_lists[r] = (covs_r[start:start + int(mine.sum())].astype(np.uint32), mine)
    cov_off = np.zeros(n_total + 1, dtype=np.uint64)
    cov_off[1:] = np.cumsum(contain_count.astype(np.uint64))
    covs = np.zeros(int(cov_off[-1]), dtype=np.uint32)
    for r in range(world):
        seg, cnts = cov_lists[r]
        if len(seg) == 0:
            continue
        local_off = np.zeros(len(cnts) + 1, dtype=np.int64)
        local_off[1:] = np.cumsum(cnts)
        gids = np.asarray(rank_genomes[r])
        nz = np.nonzero(cnts)[0]
        # scatter each genome's (already sorted) cov run to its global slot
        dst = np.repeat(cov_off[gids[nz]].astype(np.int64), cnts[nz]) + (np.arange(len(seg)) - np.repeat(local_off[nz], cnts[nz]))
        covs[dst] = seg
    return dict(contain_count=contain_count, cov_off=cov_off, covs=covs)


_PIN = {}


def gather_counts(group, contain_count, device):
    """Replicated-database mode: every rank profiled its own sample against the whole index; ONE all-gather makes the
    per-sample containment counts [world, n_genomes] available on every rank (fixed size, latency-bound)."""
    n = len(contain_count)
    key = (n, str(device))
    if key not in _PIN:
        _PIN[key] = (torch.empty(n, dtype=torch.int32, pin_memory=torch.cuda.is_available()), torch.empty(n, dtype=torch.int32, device=device),
                     torch.empty(group.world * n, dtype=torch.int32, device=device))
    host, dev, out = _PIN[key]
    host.numpy()[:] = np.asarray(contain_count).view(np.int32)
    dev.copy_(host, non_blocking=True)
    if hasattr(group, "all_gather_flat"):
        return group.all_gather_flat(dev, out).view(group.world, n)
    return torch.stack(group.all_gather_fixed(dev))


def profile_step(db, group, dk_ptr, dc_ptr, n, mine, n_total, device, _cache={}):
    """bench.py glue: db is a sylph_amd.Database holding this rank's shard; (dk_ptr, dc_ptr, n) the device-resident
    sample table of this rank."""
    if group.world == 1:
        cc, off, covs = db.contain_view(dk_ptr, dc_ptr, device_ptrs=True, n=n, packed=True)   # borrowed pinned views
        return dict(contain_count=cc, cov_off=off, covs=covs, n_occurrences=None)
    key = (id(db), n_total)
    if key not in _cache:
        sizes = group.all_gather_var(torch.from_numpy(np.asarray(mine, dtype=np.int64)).to(device))
        rank_genomes = [to_host_numpy(t).copy() for t in sizes]
        owner = np.zeros(n_total, dtype=np.int32)
        for r, g in enumerate(rank_genomes):
            owner[g] = r
        _cache[key] = (owner, rank_genomes)
    owner, rank_genomes = _cache[key]
    sk = device_view(dk_ptr, n, torch.int64, device)
    sc = device_view(dc_ptr, n, torch.int32, device)

    def contain_fn(k, c):   # borrowed pinned views; exchange_and_profile copies what it keeps before the next call
        return db.contain_view(k.data_ptr(), c.data_ptr(), device_ptrs=True, n=k.numel())

    res = exchange_and_profile(contain_fn, group, sk, sc, owner, rank_genomes)
    res["n_occurrences"] = None
    return res
