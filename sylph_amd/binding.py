"""ctypes binding of include/sylph_hip.h.  One class per handle type; method names mirror the reference functions
each ABI entry point replaces (see the header for file:line citations).  No computation happens in Python."""
import ctypes as C
import os

import numpy as np

SEED_SCALAR, SEED_AVX2_COMPAT = 0, 1
READS_SINGLE, READS_PAIRED = 0, 1
MEM_HOST, MEM_DEVICE, MEM_HOST_PINNED = 0, 1, 2
ENC_ASCII, ENC_2BIT = 0, 1

_LIB = None


class SylphHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sylph_hip error {code}: {msg}")
        self.code = code


def lib_path():
    if os.environ.get("SYLPH_HIP_LIBRARY"):           # A/B builds of the same ABI (tools/build_variant.sh)
        return os.path.abspath(os.environ["SYLPH_HIP_LIBRARY"])
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsylph_hip.so")


EXPORTS = ["sylph_version", "sylph_last_error", "sylph_free", "sylph_pinned_alloc", "sylph_pinned_free", "sylph_ctx_create", "sylph_ctx_destroy",
           "sylph_ctx_synchronize", "sylph_ctx_set_option", "sylph_ctx_profile", "sylph_ctx_kernel_stats", "sylph_seeds",
           "sylph_seeds_positions", "sylph_sketch_genome", "sylph_sketch_genomes", "sylph_sketch_begin", "sylph_sketch_push", "sylph_sketch_push_n",
           "sylph_sketch_finish", "sylph_sketch_finish_device", "sylph_sketch_destroy", "sylph_db_upload",
           "sylph_db_n_genomes", "sylph_db_n_kmers", "sylph_db_contain", "sylph_db_contain_view", "sylph_db_contain_view_packed", "sylph_db_attach_tracked", "sylph_db_reassign_view", "sylph_db_destroy",
           "sylph_sketch_push_enc", "sylph_pack_2bit", "sylph_db_index_bytes", "sylph_db_contain_batch", "sylph_shard_bounds", "sylph_db_upload_shard", "sylph_comm_rccl_unique_id",
           "sylph_comm_create_rccl", "sylph_comm_create", "sylph_comm_destroy", "sylph_db_contain_batch_sharded",
           "sylph_pipeline_create", "sylph_pipeline_submit", "sylph_pipeline_submit_session", "sylph_pipeline_flush", "sylph_pipeline_next",
           "sylph_pipeline_outstanding", "sylph_pipeline_set_option", "sylph_pipeline_profile", "sylph_pipeline_kernel_stats",
           "sylph_pipeline_destroy", "sylph_db_exchange_stats", "sylph_sketch_set_option",
           "sylph_upload_begin", "sylph_upload_chunk", "sylph_upload_commit", "sylph_upload_finish", "sylph_upload_restart", "sylph_upload_destroy",
           "sylph_db_replicate", "sylph_pipeline_create_multi", "sylph_pipeline_replica_of_last", "sylph_device_count",
           "sylph_genome_shard_bounds", "sylph_db_upload_genome_shard",
           "sylph_fastq_index", "sylph_fastq_counts", "sylph_fastq_lengths", "sylph_sketch_push_fastq", "sylph_fastq_destroy",
           "sylph_inflate", "sylph_inflate_files", "sylph_inflated_file", "sylph_inflated_text", "sylph_inflated_info", "sylph_inflated_read", "sylph_inflated_destroy"]


def load():
    """Load libsylph_hip.so.  Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise SylphHipError(-100, f"{p} is missing: build it with `make -C sylph_amd/csrc` (no CPU fallback exists)")
    L = C.CDLL(p)
    vp, u64, u32, i32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_double
    P = C.POINTER
    L.sylph_version.restype = i32
    L.sylph_last_error.restype = C.c_char_p
    L.sylph_free.argtypes = [vp]
    L.sylph_free.restype = None
    L.sylph_pinned_alloc.argtypes = [u64, P(vp)]
    L.sylph_pinned_free.argtypes = [vp]
    L.sylph_pinned_free.restype = None
    L.sylph_ctx_create.argtypes = [i32, vp, P(vp)]
    L.sylph_ctx_destroy.argtypes = [vp]
    L.sylph_ctx_destroy.restype = None
    L.sylph_ctx_synchronize.argtypes = [vp]
    L.sylph_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.sylph_ctx_profile.argtypes = [vp, i32]
    L.sylph_ctx_kernel_stats.argtypes = [vp, C.c_char_p, P(dbl), P(u64)]
    L.sylph_seeds.argtypes = [vp, vp, u64, u32, u32, i32, P(vp), P(u64)]
    L.sylph_seeds_positions.argtypes = [vp, vp, vp, u64, u32, u32, i32, P(vp), P(vp), P(vp), P(u64)]
    L.sylph_sketch_genome.argtypes = [vp, vp, vp, u64, u32, u32, i32, u64, i32, P(vp), P(u64), P(vp), P(u64)]
    L.sylph_sketch_genomes.argtypes = [vp, vp, vp, u64, vp, u64, u32, u32, i32, u64, i32, i32, P(vp), vp, P(vp), vp]
    L.sylph_sketch_begin.argtypes = [vp, u32, u32, i32, i32, i32, P(vp)]
    L.sylph_sketch_push.argtypes = [vp, vp, vp, u64, i32]
    L.sylph_sketch_push_n.argtypes = [vp, vp, vp, u64, u64, i32]
    L.sylph_sketch_push_enc.argtypes = [vp, vp, vp, u64, u64, i32, i32]
    L.sylph_fastq_index.argtypes = [vp, vp, u64, i32, P(vp)]
    L.sylph_fastq_counts.argtypes = [vp, P(u64), P(u64)]
    L.sylph_fastq_lengths.argtypes = [vp, u64, u64, vp]
    L.sylph_sketch_push_fastq.argtypes = [vp, vp, vp, u64, u64]
    L.sylph_fastq_destroy.argtypes = [vp]
    L.sylph_fastq_destroy.restype = None
    L.sylph_inflate.argtypes = [vp, vp, u64, i32, P(vp)]
    L.sylph_inflated_text.argtypes = [vp, P(vp), P(u64)]
    L.sylph_inflate_files.argtypes = [vp, P(vp), P(u64), C.c_uint32, i32, P(vp)]
    L.sylph_inflated_file.argtypes = [vp, C.c_uint32, P(vp), P(u64)]
    L.sylph_inflated_info.argtypes = [vp, P(u64), P(u64), P(u64), P(u64), P(u64)]
    L.sylph_inflated_read.argtypes = [vp, u64, u64, vp]
    L.sylph_inflated_destroy.argtypes = [vp]
    L.sylph_inflated_destroy.restype = None
    L.sylph_pack_2bit.argtypes = [vp, u64, vp]
    L.sylph_sketch_finish.argtypes = [vp, P(vp), P(vp), P(u64), P(u64)]
    L.sylph_sketch_finish_device.argtypes = [vp, P(vp), P(vp), P(u64), P(u64)]
    L.sylph_sketch_destroy.argtypes = [vp]
    L.sylph_sketch_destroy.restype = None
    L.sylph_db_upload.argtypes = [vp, vp, vp, u64, i32, P(vp)]
    L.sylph_db_n_genomes.argtypes = [vp]
    L.sylph_db_n_genomes.restype = u64
    L.sylph_db_n_kmers.argtypes = [vp]
    L.sylph_db_n_kmers.restype = u64
    L.sylph_db_contain.argtypes = [vp, vp, vp, u64, i32, dbl, vp, vp, P(vp)]
    L.sylph_db_contain_view.argtypes = [vp, vp, vp, u64, i32, dbl, P(vp), P(vp), P(vp), P(u64)]
    L.sylph_db_contain_view_packed.argtypes = [vp, vp, vp, u64, i32, dbl, P(vp), P(vp), P(vp), P(u32), P(u64)]
    L.sylph_db_attach_tracked.argtypes = [vp, vp, vp, i32]
    L.sylph_db_reassign_view.argtypes = [vp, vp, vp, u64, i32, vp, vp, u32, P(vp), P(vp), P(vp), P(u64), P(vp)]
    L.sylph_db_destroy.argtypes = [vp]
    L.sylph_db_destroy.restype = None
    L.sylph_db_index_bytes.argtypes = [vp]
    L.sylph_db_index_bytes.restype = u64
    L.sylph_db_contain_batch.argtypes = [vp, vp, u32, i32, dbl, P(vp), P(vp), P(vp), P(u32), P(u64)]
    L.sylph_shard_bounds.argtypes = [u64, u32, vp]
    L.sylph_db_upload_shard.argtypes = [vp, vp, vp, u64, i32, vp, u32, u32, P(vp)]
    L.sylph_comm_rccl_unique_id.argtypes = [vp]
    L.sylph_comm_create_rccl.argtypes = [vp, u32, u32, vp, P(vp)]
    L.sylph_comm_create.argtypes = [u32, u32, vp, vp, P(vp)]
    L.sylph_comm_destroy.argtypes = [vp]
    L.sylph_comm_destroy.restype = None
    L.sylph_db_contain_batch_sharded.argtypes = [vp, vp, vp, u32, i32, dbl, P(vp), P(vp), P(vp), P(u32), P(u64)]
    L.sylph_db_exchange_stats.argtypes = [vp, P(u64), P(u64), P(u64), i32]
    L.sylph_pipeline_create.argtypes = [vp, vp, P(vp)]
    L.sylph_pipeline_submit.argtypes = [vp, vp, u32, i32, i32, u64]
    L.sylph_pipeline_submit_session.argtypes = [vp, vp, u64]
    L.sylph_pipeline_flush.argtypes = [vp]
    L.sylph_pipeline_next.argtypes = [vp, vp]
    L.sylph_pipeline_outstanding.argtypes = [vp]
    L.sylph_pipeline_outstanding.restype = u32
    L.sylph_pipeline_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.sylph_pipeline_profile.argtypes = [vp, i32]
    L.sylph_pipeline_kernel_stats.argtypes = [vp, C.c_char_p, P(dbl), P(u64)]
    L.sylph_pipeline_destroy.argtypes = [vp]
    L.sylph_pipeline_destroy.restype = None
    L.sylph_db_replicate.argtypes = [vp, vp, P(vp)]
    L.sylph_pipeline_create_multi.argtypes = [vp, u32, vp, P(vp)]
    L.sylph_pipeline_replica_of_last.argtypes = [vp]
    L.sylph_device_count.restype = i32
    L.sylph_genome_shard_bounds.argtypes = [vp, u64, u32, vp]
    L.sylph_db_upload_genome_shard.argtypes = [vp, vp, vp, u64, i32, vp, u32, u32, P(vp)]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise SylphHipError(rc, load().sylph_last_error().decode("utf-8", "replace"))


ERR_FORMAT = -5


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _take(ptr, n, dtype):
    """Copy a library-allocated array into numpy and free it."""
    n = int(n)
    out = np.empty(n, dtype=dtype)
    if n:
        C.memmove(out.ctypes.data, ptr, n * out.itemsize)
    load().sylph_free(ptr)
    return out


def _bases(b):
    if isinstance(b, (bytes, bytearray)):
        return np.frombuffer(bytes(b), dtype=np.uint8)
    return _np(b, np.uint8)


class Context:
    """One GPU + one HIP stream.  stream: int handle of a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=-1, stream=None):
        self._h = C.c_void_p()
        _check(load().sylph_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self._h)))

    def close(self):
        if self._h:
            load().sylph_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(load().sylph_ctx_synchronize(self._h))

    def set_option(self, key, value):
        _check(load().sylph_ctx_set_option(self._h, key.encode(), value.encode()))

    def profile(self, enable=True):
        _check(load().sylph_ctx_profile(self._h, int(enable)))

    def kernel_stats(self, family):
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(load().sylph_ctx_kernel_stats(self._h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # extract_markers, sketch.rs:53
    def extract_markers(self, seq, c=200, k=31, seed_mode=SEED_AVX2_COMPAT):
        a = _bases(seq)
        out, n = C.c_void_p(), C.c_uint64(0)
        _check(load().sylph_seeds(self._h, _ptr(a) if len(a) else None, len(a), c, k, seed_mode, C.byref(out), C.byref(n)))
        return _take(out, n.value, np.uint64)

    # extract_markers_positions for all contigs, sketch.rs:71 (+ vec.sort(), sketch.rs:593)
    def extract_markers_positions(self, bases, contig_off, c=200, k=31, seed_mode=SEED_AVX2_COMPAT):
        a, off = _bases(bases), _np(contig_off, np.uint64)
        oc, op, oh, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        _check(load().sylph_seeds_positions(self._h, _ptr(a) if len(a) else None, _ptr(off), len(off) - 1, c, k, seed_mode,
                                            C.byref(oc), C.byref(op), C.byref(oh), C.byref(n)))
        return _take(oc, n.value, np.uint32), _take(op, n.value, np.uint64), _take(oh, n.value, np.uint64)

    # sketch_genome, sketch.rs:550
    def sketch_genome(self, bases, contig_off, c=200, k=31, seed_mode=SEED_AVX2_COMPAT, min_spacing=30, pseudotax=True):
        a, off = _bases(bases), _np(contig_off, np.uint64)
        ok, ot, n, nt = C.c_void_p(), C.c_void_p(), C.c_uint64(0), C.c_uint64(0)
        _check(load().sylph_sketch_genome(self._h, _ptr(a) if len(a) else None, _ptr(off), len(off) - 1, c, k, seed_mode,
                                          min_spacing, int(pseudotax), C.byref(ok), C.byref(n), C.byref(ot), C.byref(nt)))
        return dict(genome_kmers=_take(ok, n.value, np.uint64), tracked=_take(ot, nt.value, np.uint64),
                    gn_size=int(off[-1]) if len(off) else 0)


    # a batch of genomes (database build, sketch.rs:422-476): dup removal + spacing on the device
    def sketch_genomes(self, bases, contig_off, genome_contig_off, c=200, k=31, seed_mode=SEED_AVX2_COMPAT, min_spacing=30,
                       pseudotax=True, device_ptr=None):
        """-> (kmers, kmer_off, tracked, tracked_off).  `bases` is a host array, or pass device_ptr (int) with bases=None."""
        off, goff = _np(contig_off, np.uint64), _np(genome_contig_off, np.uint64)
        G = len(goff) - 1
        koff, toff = np.zeros(G + 1, dtype=np.uint64), np.zeros(G + 1, dtype=np.uint64)
        ok, ot = C.c_void_p(), C.c_void_p()
        if device_ptr is None:
            a = _bases(bases)
            ptr, mem = (_ptr(a) if len(a) else None), MEM_HOST
        else:
            ptr, mem = C.c_void_p(int(device_ptr)), MEM_DEVICE
        _check(load().sylph_sketch_genomes(self._h, ptr, _ptr(off), len(off) - 1, _ptr(goff), G, c, k, seed_mode, min_spacing,
                                           int(pseudotax), mem, C.byref(ok), _ptr(koff), C.byref(ot), _ptr(toff)))
        return _take(ok, int(koff[-1]), np.uint64), koff, _take(ot, int(toff[-1]), np.uint64), toff


def pack_2bit(ascii_bases):
    """sylph_pack_2bit: BYTE_TO_SEQ codes, 4 bases per byte, first base in the top bits (host function, no GPU needed)."""
    a = _bases(ascii_bases)
    out = np.zeros((len(a) + 3) // 4 + 16, dtype=np.uint8)
    _check(load().sylph_pack_2bit(_ptr(a) if len(a) else None, len(a), _ptr(out)))
    return out


class PinnedBuffer:
    """Page-locked host memory (sylph_pinned_alloc) exposed as a numpy array, for SYLPH_MEM_HOST_PINNED pushes."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(load().sylph_pinned_alloc(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.ptr))

    def close(self):
        if self.ptr:
            self.array = None
            load().sylph_pinned_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Inflated:
    """sylph_inflate: the bytes of one gzip file (bytes / uint8 array, host memory) inflated on the device (csrc/inflate.hip).  The text
    stays in HBM: .dev_ptr / .n_bytes go to FastqText(ctx, ptr, MEM_DEVICE, n_bytes); read() copies a range back.  Raises SylphHipError
    with code ERR_FORMAT when the library declines the stream (not gzip, damaged, ...): inflate on the host then."""

    def __init__(self, ctx, gz):
        """gz: the bytes of one file, or a list of them (sylph_inflate_files: one pass over all; .files[i] = (dev_ptr, n_bytes))"""
        self._h = None
        many = isinstance(gz, (list, tuple))
        keeps = [np.frombuffer(g, dtype=np.uint8) if isinstance(g, (bytes, bytearray, memoryview)) else _np(g, np.uint8) for g in (gz if many else [gz])]
        h = C.c_void_p()
        if many:
            ptrs = (C.c_void_p * len(keeps))(*[k.ctypes.data for k in keeps])
            lens = (C.c_uint64 * len(keeps))(*[len(k) for k in keeps])
            _check(load().sylph_inflate_files(ctx._h, ptrs, lens, len(keeps), MEM_HOST, C.byref(h)))
        else:
            _check(load().sylph_inflate(ctx._h, _ptr(keeps[0]) if len(keeps[0]) else None, len(keeps[0]), MEM_HOST, C.byref(h)))
        self._h = h
        self.files = []
        for i in range(len(keeps)):
            fp, fn = C.c_void_p(), C.c_uint64(0)
            _check(load().sylph_inflated_file(self._h, i, C.byref(fp), C.byref(fn)))
            self.files.append((int(fp.value or 0), int(fn.value)))
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(load().sylph_inflated_text(self._h, C.byref(p), C.byref(n)))
        self.dev_ptr, self.n_bytes = int(p.value or 0), int(n.value)
        a, b, c, d, e = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(load().sylph_inflated_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        self.n_members, self.n_blocks, self.n_candidates, self.n_host_members, self.n_decoded_again = (int(x.value) for x in (a, b, c, d, e))

    def read(self, first=0, n=None):
        n = self.n_bytes - first if n is None else n
        out = np.zeros(n, dtype=np.uint8)
        _check(load().sylph_inflated_read(self._h, int(first), int(n), _ptr(out)))
        return out

    def close(self):
        if self._h:
            load().sylph_inflated_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastqText:
    """sylph_fastq_*: plain four-line FASTQ text whose records the DEVICE finds (csrc/fastq.hip).  `text`: bytes / a uint8 array
    (MEM_HOST), or an integer address with n_bytes (MEM_HOST_PINNED / MEM_DEVICE; device text is borrowed until close()).
    Raises SylphHipError with code ERR_FORMAT when the text is not exactly that: parse it on the host then."""

    def __init__(self, ctx, text, mem=MEM_HOST, n_bytes=None):
        self._h = None
        h = C.c_void_p()
        if mem == MEM_HOST:
            self._keep = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else _np(text, np.uint8)
            _check(load().sylph_fastq_index(ctx._h, _ptr(self._keep) if len(self._keep) else None, len(self._keep), mem, C.byref(h)))
            self._keep = None
        else:
            _check(load().sylph_fastq_index(ctx._h, C.c_void_p(int(text)), int(n_bytes), mem, C.byref(h)))
        self._h = h
        nr, nb = C.c_uint64(0), C.c_uint64(0)
        _check(load().sylph_fastq_counts(self._h, C.byref(nr), C.byref(nb)))
        self.n_records, self.n_bases = int(nr.value), int(nb.value)

    def lengths(self, first=0, n=None):
        n = self.n_records - first if n is None else n
        out = np.zeros(n, dtype=np.uint32)
        _check(load().sylph_fastq_lengths(self._h, int(first), int(n), _ptr(out)))
        return out

    def close(self):
        if self._h:
            load().sylph_fastq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadSketcher:
    """Session for one sample: sketch_sequences_needle (sketch.rs:897) / sketch_pair_sequences (sketch.rs:771).  dedup_fpr = 0: the
    exact pair set (--fpr 0, sketch.rs:690-731); > 0 (paired sessions): the reference's default, the set behind a scalable cuckoo
    filter of that false-positive probability (sketch.rs:733-769; csrc/a10.hip), dedup_capacity its initial capacity (10^7)."""

    def __init__(self, ctx, c=200, k=31, paired=False, no_dedup=False, seed_mode=SEED_AVX2_COMPAT, dedup_fpr=0.0, dedup_capacity=None, a10=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        _check(load().sylph_sketch_begin(ctx._h, c, k, READS_PAIRED if paired else READS_SINGLE, int(no_dedup), seed_mode,
                                         C.byref(self._h)))
        if dedup_fpr:
            self.set_option("dedup_fpr", repr(float(dedup_fpr)))
        if dedup_capacity is not None:
            self.set_option("dedup_capacity", int(dedup_capacity))
        if a10 is not None:            # "auto" | "walk" | "part": which pass marks the filter's answers (csrc/a10.hip; A/B and tests)
            self.set_option("a10", a10)

    def set_option(self, key, value):
        """sylph_sketch_set_option ("borrow_until_finish": device batches stay valid until finish; "dedup_fpr", "dedup_capacity", "a10")"""
        _check(load().sylph_sketch_set_option(self._h, key.encode(), str(value).encode()))

    def push(self, bases, rec_off):
        a, off = _bases(bases), _np(rec_off, np.uint64)
        _check(load().sylph_sketch_push(self._h, _ptr(a) if len(a) else None, _ptr(off), len(off) - 1, MEM_HOST))

    def push_pinned(self, buf, bases_len, rec_off_view):
        """Batch living in page-locked memory from pinned_alloc(): buf (PinnedBuffer) holds the bases in [0, bases_len),
        rec_off_view is a uint64 numpy view of another PinnedBuffer (n_records + 1 entries)."""
        off = rec_off_view
        _check(load().sylph_sketch_push(self._h, C.c_void_p(buf.ptr), C.c_void_p(off.ctypes.data), len(off) - 1, MEM_HOST_PINNED))

    def push_device(self, bases_ptr, rec_off_ptr, n_records, n_bases=None):
        """bases_ptr / rec_off_ptr: integer device addresses (e.g. torch tensor .data_ptr()); n_bases = rec_off[n_records]
        if the caller knows it (saves a device->host read)."""
        if n_bases is None:
            _check(load().sylph_sketch_push(self._h, C.c_void_p(bases_ptr), C.c_void_p(rec_off_ptr), n_records, MEM_DEVICE))
        else:
            _check(load().sylph_sketch_push_n(self._h, C.c_void_p(bases_ptr), C.c_void_p(rec_off_ptr), n_records, n_bases,
                                              MEM_DEVICE))

    def push_enc(self, bases, rec_off, n_bases, mem=MEM_HOST, enc=ENC_ASCII, n_records=None):
        """sylph_sketch_push_enc.  bases / rec_off: numpy arrays (MEM_HOST), integer addresses of page-locked memory
        (MEM_HOST_PINNED) or of device memory (MEM_DEVICE; then n_records must be given)."""
        if mem == MEM_HOST:
            a, off = _np(bases, np.uint8), _np(rec_off, np.uint64)
            _check(load().sylph_sketch_push_enc(self._h, _ptr(a) if len(a) else None, _ptr(off), len(off) - 1, int(n_bases), mem, enc))
        else:
            _check(load().sylph_sketch_push_enc(self._h, C.c_void_p(int(bases)), C.c_void_p(int(rec_off)), int(n_records), int(n_bases), mem, enc))

    def push_fastq(self, a, b=None, first=0, n_items=None):
        """sylph_sketch_push_fastq: records [first, first + n_items) of the FastqText `a` (and of `b`, the mates, in a paired session)."""
        if n_items is None:
            n_items = (min(a.n_records, b.n_records) if b is not None else a.n_records) - first
        _check(load().sylph_sketch_push_fastq(self._h, a._h, b._h if b is not None else None, int(first), int(n_items)))

    def finish(self):
        ok, oc, n, d = C.c_void_p(), C.c_void_p(), C.c_uint64(0), C.c_uint64(0)
        _check(load().sylph_sketch_finish(self._h, C.byref(ok), C.byref(oc), C.byref(n), C.byref(d)))
        return dict(kmers=_take(ok, n.value, np.uint64), counts=_take(oc, n.value, np.uint32), dup_removed=int(d.value))

    def finish_device(self):
        """-> (device address of kmers, device address of counts, n, dup_removed); valid until close()."""
        ok, oc, n, d = C.c_void_p(), C.c_void_p(), C.c_uint64(0), C.c_uint64(0)
        _check(load().sylph_sketch_finish_device(self._h, C.byref(ok), C.byref(oc), C.byref(n), C.byref(d)))
        return ok.value or 0, oc.value or 0, int(n.value), int(d.value)

    def close(self):
        if self._h:
            load().sylph_sketch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SampleRef(C.Structure):
    _fields_ = [("kmers", C.c_void_p), ("counts", C.c_void_p), ("n", C.c_uint64)]


def shard_bounds(max_kmer, world):
    b = np.zeros(world + 1, dtype=np.uint64)
    _check(load().sylph_shard_bounds(int(max_kmer), world, _ptr(b)))
    return b


class CommOps(C.Structure):
    _fields_ = [("all_gather", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)),
                ("all_to_all", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p))]


class Comm:
    """sylph_comm: RCCL (rccl_id = the 128-byte id from Comm.rccl_unique_id(), the same on every rank) or caller-supplied
    collectives: all_gather(send_ptr, recv_ptr, nbytes, stream) / all_to_all(send_ptr, send_off, recv_ptr, recv_off, stream)
    working on device addresses."""

    @staticmethod
    def rccl_unique_id():
        buf = (C.c_uint8 * 128)()
        _check(load().sylph_comm_rccl_unique_id(buf))
        return bytes(buf)

    def __init__(self, rank, world, ctx=None, rccl_id=None, all_gather=None, all_to_all=None):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        if rccl_id is not None:
            buf = (C.c_uint8 * 128).from_buffer_copy(rccl_id)
            _check(load().sylph_comm_create_rccl(ctx._h, rank, world, buf, C.byref(self._h)))
        else:
            def ag(user, send, recv, nbytes, stream):
                try:
                    all_gather(send, recv, nbytes, stream)
                    return 0
                except Exception as e:   # never unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1

            def a2a(user, send, send_off, recv, recv_off, stream):
                try:
                    all_to_all(send, [send_off[i] for i in range(world + 1)], recv, [recv_off[i] for i in range(world + 1)], stream)
                    return 0
                except Exception as e:
                    import traceback
                    traceback.print_exc()
                    return 1
            self._ops = CommOps(CommOps._fields_[0][1](ag), CommOps._fields_[1][1](a2a))   # keep the thunks alive
            _check(load().sylph_comm_create(rank, world, C.byref(self._ops), None, C.byref(self._h)))

    def close(self):
        if self._h:
            load().sylph_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Database:
    """genome_kmers of many GenomeSketch resident in HBM + postings index; probe half of get_stats (contain.rs:601-656)."""

    def __init__(self, ctx, kmers, genome_off, device_ptrs=False, n_genomes=None, shard=None, genome_shard=None):
        """shard = (bounds[world + 1], world, rank): keep only the k-mers of this rank's range (sylph_db_upload_shard);
        genome_shard = (g_bounds[world + 1], world, rank): keep the genomes [g_bounds[rank], g_bounds[rank + 1]) (sylph_db_upload_genome_shard)."""
        self.ctx = ctx
        self._h = C.c_void_p()
        if device_ptrs:
            kp, op, G, mem = C.c_void_p(kmers), C.c_void_p(genome_off), n_genomes, MEM_DEVICE
        else:
            k, off = _np(kmers, np.uint64), _np(genome_off, np.uint64)
            kp, op, G, mem = (_ptr(k) if len(k) else None), _ptr(off), len(off) - 1, MEM_HOST
        if genome_shard is not None:
            gb = _np(genome_shard[0], np.uint64)
            assert len(gb) == genome_shard[1] + 1
            _check(load().sylph_db_upload_genome_shard(ctx._h, kp, op, G, mem, _ptr(gb), genome_shard[1], genome_shard[2], C.byref(self._h)))
        elif shard is None:
            _check(load().sylph_db_upload(ctx._h, kp, op, G, mem, C.byref(self._h)))
        else:
            b = _np(shard[0], np.uint64)
            assert len(b) == shard[1] + 1
            _check(load().sylph_db_upload_shard(ctx._h, kp, op, G, mem, _ptr(b), shard[1], shard[2], C.byref(self._h)))
        self.n_genomes = int(load().sylph_db_n_genomes(self._h))
        self.n_kmers = int(load().sylph_db_n_kmers(self._h))

    def replicate(self, ctx):
        """sylph_db_replicate: a copy of this (unsharded) database on another context — another GPU of the node; the index travels device
        to device (xGMI), nothing is uploaded or built again."""
        other = Database.__new__(Database)
        other.ctx = ctx
        other._h = C.c_void_p()
        _check(load().sylph_db_replicate(self._h, ctx._h, C.byref(other._h)))
        other.n_genomes = int(load().sylph_db_n_genomes(other._h))
        other.n_kmers = int(load().sylph_db_n_kmers(other._h))
        return other

    def contain(self, sample_kmers, sample_counts, min_number_kmers=50.0, device_ptrs=False, n=None):
        """-> (contain_count[G] uint32, cov_off[G+1] uint64, covs uint32 sorted ascending per genome)."""
        G = self.n_genomes
        cc = np.zeros(max(G, 1), dtype=np.uint32)
        off = np.zeros(G + 1, dtype=np.uint64)
        out = C.c_void_p()
        if device_ptrs:
            _check(load().sylph_db_contain(self._h, C.c_void_p(sample_kmers), C.c_void_p(sample_counts), n, MEM_DEVICE,
                                           float(min_number_kmers), _ptr(cc), _ptr(off), C.byref(out)))
        else:
            k, c = _np(sample_kmers, np.uint64), _np(sample_counts, np.uint32)
            _check(load().sylph_db_contain(self._h, _ptr(k) if len(k) else None, _ptr(c) if len(c) else None, len(k),
                                           MEM_HOST, float(min_number_kmers), _ptr(cc), _ptr(off), C.byref(out)))
        covs = _take(out, int(off[G]), np.uint32)
        return cc[:G], off, covs

    def contain_view(self, sample_kmers, sample_counts, min_number_kmers=50.0, device_ptrs=False, n=None, packed=False):
        """Zero-copy variant: numpy views of the db-owned pinned result buffers, valid until the next contain* call.
        packed=True: coverage values come back as uint8/uint16/uint32, whichever holds the sample's largest count."""
        G = self.n_genomes
        pc, po, pv, nh, cw = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(0), C.c_uint32(4)
        if device_ptrs:
            args = (self._h, C.c_void_p(sample_kmers), C.c_void_p(sample_counts), n, MEM_DEVICE, float(min_number_kmers))
        else:
            k, c = _np(sample_kmers, np.uint64), _np(sample_counts, np.uint32)
            args = (self._h, _ptr(k) if len(k) else None, _ptr(c) if len(c) else None, len(k), MEM_HOST, float(min_number_kmers))
        if packed:
            _check(load().sylph_db_contain_view_packed(*args, C.byref(pc), C.byref(po), C.byref(pv), C.byref(cw), C.byref(nh)))
        else:
            _check(load().sylph_db_contain_view(*args, C.byref(pc), C.byref(po), C.byref(pv), C.byref(nh)))

        def view(ptr, count, ctype, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).view(dtype)
        off = view(po, G + 1, C.c_uint64, np.uint64)
        cc = view(pc, G, C.c_uint32, np.uint32)
        ct, dt = {1: (C.c_uint8, np.uint8), 2: (C.c_uint16, np.uint16), 4: (C.c_uint32, np.uint32)}[int(cw.value)]
        covs = view(pv, int(nh.value), ct, dt)
        return cc, off, covs

    @property
    def index_bytes(self):
        return int(load().sylph_db_index_bytes(self._h))

    def _batch_args(self, samples, device_ptrs):
        """samples: list of (kmers, counts) numpy arrays, or of (kmers_ptr, counts_ptr, n) device addresses."""
        arr = (SampleRef * max(1, len(samples)))()
        keep = []
        for i, smp in enumerate(samples):
            if device_ptrs:
                arr[i].kmers, arr[i].counts, arr[i].n = smp[0], smp[1], smp[2]
            else:
                k, c = _np(smp[0], np.uint64), _np(smp[1], np.uint32)
                keep.append((k, c))
                arr[i].kmers, arr[i].counts, arr[i].n = (k.ctypes.data if len(k) else None), (c.ctypes.data if len(c) else None), len(k)
        return arr, keep

    def _batch_views(self, S, pc, po, pv, nh, cw):
        R = S * self.n_genomes

        def view(ptr, count, ctype, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).view(dtype)
        ct, dt = {1: (C.c_uint8, np.uint8), 2: (C.c_uint16, np.uint16), 4: (C.c_uint32, np.uint32)}[int(cw.value)]
        return view(pc, R, C.c_uint32, np.uint32), view(po, R + 1, C.c_uint64, np.uint64), view(pv, int(nh.value), ct, dt)

    def contain_batch(self, samples, min_number_kmers=50.0, device_ptrs=False):
        """sylph_db_contain_batch -> (contain_count[S*G], cov_off[S*G+1], covs): borrowed views, row = s * G + g."""
        arr, keep = self._batch_args(samples, device_ptrs)
        pc, po, pv, nh, cw = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(0), C.c_uint32(4)
        _check(load().sylph_db_contain_batch(self._h, arr, len(samples), MEM_DEVICE if device_ptrs else MEM_HOST, float(min_number_kmers),
                                             C.byref(pc), C.byref(po), C.byref(pv), C.byref(cw), C.byref(nh)))
        return self._batch_views(len(samples), pc, po, pv, nh, cw)

    def contain_batch_sharded(self, comm, samples, min_number_kmers=50.0, device_ptrs=False):
        """sylph_db_contain_batch_sharded: collective — every rank calls it with its own samples."""
        arr, keep = self._batch_args(samples, device_ptrs)
        pc, po, pv, nh, cw = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(0), C.c_uint32(4)
        _check(load().sylph_db_contain_batch_sharded(self._h, comm._h, arr, len(samples), MEM_DEVICE if device_ptrs else MEM_HOST,
                                                     float(min_number_kmers), C.byref(pc), C.byref(po), C.byref(pv), C.byref(cw), C.byref(nh)))
        return self._batch_views(len(samples), pc, po, pv, nh, cw)

    def exchange_stats(self, reset=False):
        """-> (batches, table bytes sent to other ranks, hit bytes sent to other ranks) of a sharded database"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        _check(load().sylph_db_exchange_stats(self._h, C.byref(a), C.byref(b), C.byref(c), int(reset)))
        return int(a.value), int(b.value), int(c.value)

    def attach_tracked(self, tracked_kmers, tracked_off):
        k, off = _np(tracked_kmers, np.uint64), _np(tracked_off, np.uint64)
        _check(load().sylph_db_attach_tracked(self._h, _ptr(k) if len(k) else None, _ptr(off), MEM_HOST))

    def reassign_view(self, sample_kmers, sample_counts, passing_gids, passing_ani):
        """winner_table + second probe pass on device -> (contain_count, cov_off, covs, kmers_lost), borrowed views."""
        G = self.n_genomes
        k, c = _np(sample_kmers, np.uint64), _np(sample_counts, np.uint32)
        pg, pa = _np(passing_gids, np.uint32), _np(passing_ani, np.float64)
        pc, po, pv, pl, nh = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        _check(load().sylph_db_reassign_view(self._h, _ptr(k) if len(k) else None, _ptr(c) if len(c) else None, len(k), MEM_HOST,
                                             _ptr(pg) if len(pg) else None, _ptr(pa) if len(pa) else None, len(pg), C.byref(pc),
                                             C.byref(po), C.byref(pv), C.byref(nh), C.byref(pl)))

        def view(ptr, count, ctype, dtype):
            if count == 0:
                return np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).view(dtype)
        return (view(pc, G, C.c_uint32, np.uint32), view(po, G + 1, C.c_uint64, np.uint64),
                view(pv, int(nh.value), C.c_uint32, np.uint32), view(pl, G, C.c_uint32, np.uint32))

    def close(self):
        if self._h:
            load().sylph_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReadBatch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("rec_off", C.c_void_p), ("n_records", C.c_uint64), ("n_bases", C.c_uint64)]


class PipelineConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("n_workers", C.c_uint32), ("depth", C.c_uint32), ("max_batch", C.c_uint32),
                ("c", C.c_uint32), ("k", C.c_uint32), ("reads_mode", C.c_int), ("no_dedup", C.c_int), ("seed_mode", C.c_int),
                ("want_table", C.c_int), ("min_number_kmers", C.c_double), ("comm", C.c_void_p)]


class PipelineResult(C.Structure):
    _fields_ = [("tag", C.c_uint64), ("status", C.c_int), ("error", C.c_char_p), ("n_table", C.c_uint64), ("dup_removed", C.c_uint64),
                ("dev_kmers", C.c_void_p), ("dev_counts", C.c_void_p), ("kmers", C.c_void_p), ("counts", C.c_void_p),
                ("contain_count", C.c_void_p), ("cov_off", C.c_void_p), ("covs", C.c_void_p), ("cov_width", C.c_uint32),
                ("n_covs", C.c_uint64), ("probe_batch", C.c_uint32), ("t_submit", C.c_double), ("t_sketch_begin", C.c_double),
                ("t_sketch_end", C.c_double), ("t_profile_begin", C.c_double), ("t_done", C.c_double)]


class Pipeline:
    """sylph_pipeline: samples in (device / host batches, or sessions), results out in submission order; the overlap of the
    sketch and profile stages of different samples happens on C++ threads inside the library."""

    def __init__(self, db, c=200, k=31, paired=True, no_dedup=False, seed_mode=SEED_AVX2_COMPAT, n_workers=0, depth=0, max_batch=0,
                 want_table=False, min_number_kmers=50.0, comm=None):
        # db: one Database, or a list of replicas of one (Database.replicate: one per GPU) -> sylph_pipeline_create_multi: one sample
        # loop over all of them, results in submission order
        self.dbs = list(db) if isinstance(db, (list, tuple)) else None
        self.db = self.dbs[0] if self.dbs else db
        self._h = C.c_void_p()
        cfg = PipelineConfig(C.sizeof(PipelineConfig), n_workers, depth, max_batch, c, k, READS_PAIRED if paired else READS_SINGLE,
                             int(no_dedup), seed_mode, int(want_table), float(min_number_kmers), comm._h if comm is not None else None)
        if self.dbs:
            arr = (C.c_void_p * len(self.dbs))(*[d._h for d in self.dbs])
            _check(load().sylph_pipeline_create_multi(arr, len(self.dbs), C.byref(cfg), C.byref(self._h)))
        else:
            _check(load().sylph_pipeline_create(db._h, C.byref(cfg), C.byref(self._h)))
        self._res = PipelineResult()
        self._want_table = bool(want_table)

    def submit_device(self, batches, tag=0, enc=ENC_ASCII, mem=MEM_DEVICE):
        """batches: list of (bases_ptr, rec_off_ptr, n_records, n_bases) integer addresses.  False when `depth` samples are
        outstanding already."""
        arr = (ReadBatch * max(1, len(batches)))()
        for i, b in enumerate(batches):
            arr[i].bases, arr[i].rec_off, arr[i].n_records, arr[i].n_bases = b
        rc = load().sylph_pipeline_submit(self._h, arr, len(batches), mem, enc, tag)
        if rc == -4:
            return False
        _check(rc)
        return True

    def submit_session(self, sketcher, tag=0):
        """Hands a ReadSketcher's session over (the pipeline finishes and destroys it)."""
        rc = load().sylph_pipeline_submit_session(self._h, sketcher._h, tag)
        if rc == -4:
            return False
        _check(rc)
        sketcher._h = C.c_void_p()
        return True

    def flush(self):
        _check(load().sylph_pipeline_flush(self._h))

    def next(self, views=True):
        """-> dict of the oldest outstanding sample (numpy views valid until the next call)."""
        r = self._res
        _check(load().sylph_pipeline_next(self._h, C.byref(r)))
        if r.status != 0:
            raise SylphHipError(r.status, (r.error or b"").decode("utf-8", "replace"))
        out = dict(tag=int(r.tag), n_table=int(r.n_table), dup_removed=int(r.dup_removed), dev_kmers=r.dev_kmers or 0,
                   replica=int(load().sylph_pipeline_replica_of_last(self._h)),
                   dev_counts=r.dev_counts or 0, n_covs=int(r.n_covs), probe_batch=int(r.probe_batch),
                   t=(r.t_submit, r.t_sketch_begin, r.t_sketch_end, r.t_profile_begin, r.t_done))
        if views:
            G = self.db.n_genomes

            def view(ptr, count, ctype, dtype):
                if count == 0 or not ptr:
                    return np.zeros(0, dtype=dtype)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).view(dtype)
            ct, dt = {1: (C.c_uint8, np.uint8), 2: (C.c_uint16, np.uint16), 4: (C.c_uint32, np.uint32)}[int(r.cov_width)]
            off = view(r.cov_off, G + 1, C.c_uint64, np.uint64)
            out["contain_count"] = view(r.contain_count, G, C.c_uint32, np.uint32)
            out["cov_off"] = off
            # covs is the base of the whole batch's values; cov_off indexes it
            hi = int(off[G]) if G else 0
            out["covs"] = view(r.covs, hi, ct, dt)
            if self._want_table:
                out["kmers"] = view(r.kmers, int(r.n_table), C.c_uint64, np.uint64)
                out["counts"] = view(r.counts, int(r.n_table), C.c_uint32, np.uint32)
        return out

    @property
    def outstanding(self):
        return int(load().sylph_pipeline_outstanding(self._h))

    def set_option(self, key, value):
        _check(load().sylph_pipeline_set_option(self._h, key.encode(), (repr(float(value)) if isinstance(value, float) else str(value)).encode()))

    def profile(self, enable=True):
        _check(load().sylph_pipeline_profile(self._h, int(enable)))

    def kernel_stats(self, family):
        ms, n = C.c_double(0), C.c_uint64(0)
        _check(load().sylph_pipeline_kernel_stats(self._h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self._h:
            load().sylph_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
