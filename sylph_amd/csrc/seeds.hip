// seeds.hip — FracMinHash seeding on gfx950: canonical k-mer -> mm_hash64 -> keep if hash < u64::MAX / c.
//
// Replaces the per-sequence loops of seeding.rs:86-146 (fmh_seeds) / avx2_seeding.rs:33-148
// (extract_markers_avx2) for a whole batch of records at once.  Design:
//   * the batch is ONE flat ASCII stream in HBM (records concatenated, no separators); a workgroup of 256
//     threads owns a tile of 16384 consecutive k-mer start positions and grid-strides over tiles;
//   * load phase: 16 B/lane coalesced global loads (1 KiB per wave instruction), ASCII -> 2-bit codes with the
//     exact BYTE_TO_SEQ semantics (types.rs:50-59), packed 16 bases per dword into two LDS streams: F (forward,
//     first base in the top bits) and R (complement, first base in the bottom bits);
//   * hash phase: each lane reads 6+6 dwords from LDS and produces 64 k-mers; the forward k-mer and its reverse
//     complement are funnel-shift extracts (v_alignbit) of the two streams at compile-time shifts, so there is
//     no loop-carried rolling state and no per-base LUT access;
//   * k-mers that straddle a record boundary are hashed too and rejected later (annotate kernel) — rejecting
//     1 in c survivors is cheaper than testing every position;
//   * survivors (hash, position) are staged in LDS and flushed with one global atomic per ~1000 survivors.
// Pure integer work, no MFMA.  Algorithmic HBM traffic: 1 B per base in, 12 B per survivor out.
#include "common.h"
#include "device_common.h"

namespace sylph {

namespace {

constexpr int TPB = 256;
constexpr int WPT = 4;                        // packed dwords (16 bases each) per lane
constexpr int TILE_WORDS = TPB * WPT;         // 1024
constexpr int TILE_BASES = TILE_WORDS * 16;   // 16384
constexpr int HALO_WORDS = 2;                 // k-1 <= 31 bases beyond the tile
constexpr int STAGE_CAP = 1024;               // LDS survivor staging (12 KiB)
constexpr int FLUSH_AT = 512;
constexpr int LIST_CAP = 2048;                // survivors of a tile finished cooperatively (ordered-slots kernel); 16384 / c expected

template <int S>
__device__ __forceinline__ uint64_t shr96(uint32_t hi, uint32_t mid, uint32_t lo) {
    // low 64 bits of ((hi:mid:lo) >> S), S compile-time in [0, 64)
    uint32_t l, h;
    if constexpr (S == 0) { l = lo; h = mid; }
    else if constexpr (S < 32) { l = __builtin_amdgcn_alignbit(mid, lo, S); h = __builtin_amdgcn_alignbit(hi, mid, S); }
    else if constexpr (S == 32) { l = mid; h = hi; }
    else { l = __builtin_amdgcn_alignbit(hi, mid, S - 32); h = hi >> (S - 32); }
    return ((uint64_t)h << 32) | l;
}

struct Stage {
    uint64_t hash[STAGE_CAP];
    uint32_t pos[STAGE_CAP];
};

template <int K>
struct KmerConsts {
    static constexpr uint64_t MASK = (K == 32) ? ~0ull : ((1ull << (2 * K)) - 1);
};

// canonical k-mer hash at in-word offset O (compile-time) from three consecutive dwords of each stream
template <int K, int O, int HV>
__device__ __forceinline__ uint64_t hash_at(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t r0, uint32_t r1, uint32_t r2) {
    constexpr int SF = 96 - 2 * O - 2 * K;                 // forward: top-justified big-endian stream
    const uint64_t f = shr96<SF>(f0, f1, f2) & KmerConsts<K>::MASK;
    const uint64_t r = shr96<2 * O>(r2, r1, r0) & KmerConsts<K>::MASK;   // reverse complement: little-endian stream
    const uint64_t canon = f < r ? f : r;                  // seeding.rs:134-139
    return HV ? mm_hash64_gfx950(canon) : mm_hash64(canon);
}

// same with a run-time offset (only executed for the ~1/c k-mers that passed the threshold)
template <int K>
__device__ __forceinline__ uint64_t hash_at_dyn(uint32_t o, uint32_t f0, uint32_t f1, uint32_t f2, uint32_t r0, uint32_t r1,
                                                uint32_t r2) {
    auto shr96d = [](uint32_t hi, uint32_t mid, uint32_t lo, uint32_t sh) -> uint64_t {   // sh in [0, 63]
        const uint64_t a = ((uint64_t)mid << 32) | lo;
        return sh == 0 ? a : ((a >> sh) | ((uint64_t)hi << (64 - sh)));
    };
    const uint64_t f = shr96d(f0, f1, f2, 96 - 2 * o - 2 * K) & KmerConsts<K>::MASK;
    const uint64_t r = shr96d(r2, r1, r0, 2 * o) & KmerConsts<K>::MASK;
    return mm_hash64(f < r ? f : r);
}

// 16 k-mers of one dword: accumulate "hash < threshold" into a bit mask (k-mer O -> bit 15-O) with two VALU ops per
// k-mer and no control flow; the rare survivors are re-hashed and emitted afterwards.
template <int K, int O, int HV>
struct Unroll16 {
    static __device__ __forceinline__ void run(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t r0, uint32_t r1,
                                               uint32_t r2, uint64_t thr, uint32_t& mask) {
        const uint64_t h = hash_at<K, O, HV>(f0, f1, f2, r0, r1, r2);
        // mask = 2*mask + (h < thr): v_cmp_lt_u64 sets vcc, v_addc_co_u32 folds it in (seeding.rs:142, strict <)
        asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(h), "s"(thr) : "vcc");
        if constexpr (O + 1 < 16) Unroll16<K, O + 1, HV>::run(f0, f1, f2, r0, r1, r2, thr, mask);
    }
};

template <int K>
__device__ __forceinline__ void emit_hits(uint32_t mask, uint32_t f0, uint32_t f1, uint32_t f2, uint32_t r0, uint32_t r1,
                                          uint32_t r2, uint32_t pos0, Stage& st, uint32_t* s_cnt, uint64_t* out_hash,
                                          uint32_t* out_pos, uint32_t out_cap, uint32_t* out_count) {
    while (mask) {
        const uint32_t b = 31u - (uint32_t)__clz((int)mask);
        mask &= ~(1u << b);
        const uint32_t o = 15u - b;
        const uint64_t h = hash_at_dyn<K>(o, f0, f1, f2, r0, r1, r2);
        const uint32_t slot = atomicAdd(s_cnt, 1u);
        if (slot < STAGE_CAP) {
            st.hash[slot] = h;
            st.pos[slot] = pos0 + o;
        } else {                                           // staging full (pathological repeats): straight to HBM
            const uint32_t g = atomicAdd(out_count, 1u);
            if (g < out_cap) { out_hash[g] = h; out_pos[g] = pos0 + o; }
        }
    }
}

// K1.  n_bases < 2^32.  `bases` 16-byte aligned; chunks that start at or beyond n_bases are never read.
template <int K, int HV>
__global__ __launch_bounds__(TPB) void seeds_kernel(const uint8_t* __restrict__ bases, uint32_t n_bases, uint64_t thr,
                                                    uint32_t n_tiles, uint64_t* __restrict__ out_hash,
                                                    uint32_t* __restrict__ out_pos, uint32_t out_cap,
                                                    uint32_t* __restrict__ out_count) {
    __shared__ __attribute__((aligned(16))) uint32_t sF[TILE_WORDS + 8];
    __shared__ __attribute__((aligned(16))) uint32_t sR[TILE_WORDS + 8];
    __shared__ Stage st;
    __shared__ uint32_t s_cnt, s_base;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t tile_base = (uint64_t)tile * TILE_BASES;
        // ---- load + pack ------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j <= WPT; j++) {
            const uint32_t ci = tid + j * TPB;
            if (j == WPT && tid >= HALO_WORDS) break;
            const uint64_t b0 = tile_base + (uint64_t)ci * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b0 < n_bases) v = *reinterpret_cast<const uint4*>(bases + b0);
            uint32_t F, R;
            pack16(v, F, R);
            sF[ci] = F;
            sR[ci] = R;
        }
        __syncthreads();
        // ---- hash ---------------------------------------------------------------------------------------
        {
            const uint32_t w0 = tid * WPT;
            const uint4 fa = *reinterpret_cast<const uint4*>(&sF[w0]);
            const uint2 fb = *reinterpret_cast<const uint2*>(&sF[w0 + 4]);
            const uint4 ra = *reinterpret_cast<const uint4*>(&sR[w0]);
            const uint2 rb = *reinterpret_cast<const uint2*>(&sR[w0 + 4]);
            const uint32_t fw[6] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y};
            const uint32_t rw[6] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y};
            const uint32_t p0 = (uint32_t)tile_base + w0 * 16;
#pragma unroll
            for (int j = 0; j < WPT; j++) {
                // positions at or beyond n_bases can never be valid; skip whole dwords of them (wave-uniform
                // except in the single boundary wave)
                if ((uint64_t)p0 + (uint64_t)j * 16 < n_bases) {
                    uint32_t mask = 0;
                    Unroll16<K, 0, HV>::run(fw[j], fw[j + 1], fw[j + 2], rw[j], rw[j + 1], rw[j + 2], thr, mask);
                    if (mask)
                        emit_hits<K>(mask, fw[j], fw[j + 1], fw[j + 2], rw[j], rw[j + 1], rw[j + 2], p0 + j * 16, st, &s_cnt,
                                     out_hash, out_pos, out_cap, out_count);
                }
            }
        }
        __syncthreads();
        // ---- flush staged survivors ----------------------------------------------------------------------
        const bool last = (tile + gridDim.x >= n_tiles);
        const uint32_t n = min(s_cnt, (uint32_t)STAGE_CAP);
        if (n >= FLUSH_AT || (last && n > 0)) {
            if (tid == 0) s_base = atomicAdd(out_count, n);
            __syncthreads();
            const uint32_t base = s_base;
            for (uint32_t i = tid; i < n; i += TPB) {
                const uint32_t g = base + i;
                if (g < out_cap) { out_hash[g] = st.hash[i]; out_pos[g] = st.pos[i]; }
            }
            __syncthreads();
            if (tid == 0) s_cnt = 0;
        }
        // (the __syncthreads after the next tile's pack phase orders the s_cnt reset before new survivors)
    }
}

// K1, ordered flavour (the default).  Same load / pack / hash phases, but survivors are written in ascending position
// order into a fixed-capacity slot region per tile: lanes count their hits (popcount of the four 16-bit masks), one
// workgroup exclusive scan gives every lane its offset, and the rare hits are re-hashed and stored at
// slot[tile * slot_cap + offset].  No atomics, deterministic output, and — because tiles are in position order too — a
// scan + gather over the tile counts replaces the device-wide radix sort by position that the annotate kernel needs.
// A tile that overflows its slots (pathological repeats) bumps `overflow`; the host then reruns the batch through the
// unordered kernel + radix sort.
template <int K, int HV>
__global__ __launch_bounds__(TPB) void seeds_slots_kernel(const uint8_t* __restrict__ bases, uint32_t n_bases, uint64_t thr,
                                                          uint32_t n_tiles, uint32_t slot_cap, uint64_t* __restrict__ slot_hash,
                                                          uint32_t* __restrict__ slot_pos, uint32_t* __restrict__ tile_count,
                                                          SpillState* __restrict__ spill, const uint32_t* __restrict__ tile_list,
                                                          uint32_t* __restrict__ spill_slot_of_tile) {
    __shared__ __attribute__((aligned(16))) uint32_t sF[TILE_WORDS + 8];
    __shared__ __attribute__((aligned(16))) uint32_t sR[TILE_WORDS + 8];
    __shared__ uint32_t s_wave[TPB / 64];
    __shared__ uint16_t s_list[LIST_CAP];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first pass: it = tile; redo pass (tile_list != nullptr): it = index into the list of tiles that overflowed their slots,
    // which now get a slot region of TILE_BASES entries each (n_tiles = length of the list)
    for (uint32_t it = blockIdx.x; it < n_tiles; it += gridDim.x) {
        const uint32_t tile = tile_list ? tile_list[it] : it;
        const uint64_t tile_base = (uint64_t)tile * TILE_BASES;
#pragma unroll
        for (int j = 0; j <= WPT; j++) {
            const uint32_t ci = tid + j * TPB;
            if (j == WPT && tid >= HALO_WORDS) break;
            const uint64_t b0 = tile_base + (uint64_t)ci * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (b0 < n_bases) v = *reinterpret_cast<const uint4*>(bases + b0);
            uint32_t F, R;
            pack16(v, F, R);
            sF[ci] = F;
            sR[ci] = R;
        }
        __syncthreads();
        const uint32_t w0 = tid * WPT;
        const uint4 fa = *reinterpret_cast<const uint4*>(&sF[w0]);
        const uint2 fb = *reinterpret_cast<const uint2*>(&sF[w0 + 4]);
        const uint4 ra = *reinterpret_cast<const uint4*>(&sR[w0]);
        const uint2 rb = *reinterpret_cast<const uint2*>(&sR[w0 + 4]);
        const uint32_t fw[6] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y};
        const uint32_t rw[6] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y};
        const uint32_t p0 = (uint32_t)tile_base + w0 * 16;
        uint32_t masks[WPT];
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < WPT; j++) {
            masks[j] = 0;
            if ((uint64_t)p0 + (uint64_t)j * 16 < n_bases)
                Unroll16<K, 0, HV>::run(fw[j], fw[j + 1], fw[j + 2], rw[j], rw[j + 1], rw[j + 2], thr, masks[j]);
            cnt += __popc(masks[j]);
        }
        // workgroup exclusive scan of cnt
        uint32_t x = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= (uint32_t)d) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();   // also: every lane has finished reading sF/sR of this tile
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; w++) {
            const uint32_t t = s_wave[w];
            if ((uint32_t)w < wave) base += t;
            total += t;
        }
        uint32_t o = base + x - cnt;
        const uint64_t out0 = (uint64_t)it * slot_cap;
        if (total <= (uint32_t)LIST_CAP) {
            // The tile's survivors are finished COOPERATIVELY (round 3, as in the read-per-lane kernel): the lanes that own them
            // only write a 16-bit descriptor (stream word, offset in the word) at the survivor's place in the tile's list — a
            // few instructions per hit — then the list is dealt one survivor per lane and re-hashed from the LDS streams.  With
            // every lane re-hashing its own hits, a wavefront ran the ~80-instruction body as often as its busiest lane had
            // hits: 3-4 times at c = 100 for 0.6 hits per lane.
            if (cnt) {
                uint32_t q = o;
#pragma unroll
                for (int j = 0; j < WPT; j++) {
                    uint32_t m = masks[j];
                    while (m) {   // bit 15-O <-> in-dword offset O: highest bit first = ascending position
                        const uint32_t b = 31u - (uint32_t)__clz((int)m);
                        m &= ~(1u << b);
                        s_list[q++] = (uint16_t)(((w0 + j) << 4) | (15u - b));
                    }
                }
            }
            __syncthreads();
            for (uint32_t hix = tid; hix < min(total, slot_cap); hix += TPB) {
                const uint32_t e = s_list[hix], w = e >> 4, off = e & 15u;
                slot_hash[out0 + hix] = hash_at_dyn<K>(off, sF[w], sF[w + 1], sF[w + 2], sR[w], sR[w + 1], sR[w + 2]);
                slot_pos[out0 + hix] = (uint32_t)tile_base + w * 16 + off;
            }
        } else if (cnt) {       // more survivors than the list holds (low-complexity sequence, tiny c): every lane its own
#pragma unroll
            for (int j = 0; j < WPT; j++) {
                uint32_t m = masks[j];
                while (m) {
                    const uint32_t b = 31u - (uint32_t)__clz((int)m);
                    m &= ~(1u << b);
                    const uint32_t off = 15u - b;
                    if (o < slot_cap) {
                        slot_hash[out0 + o] = hash_at_dyn<K>(off, fw[j], fw[j + 1], fw[j + 2], rw[j], rw[j + 1], rw[j + 2]);
                        slot_pos[out0 + o] = p0 + j * 16 + off;
                    }
                    o++;
                }
            }
        }
        if (tid == 0) {
            if (tile_list) {
                spill_slot_of_tile[tile] = it;
            } else {
                tile_count[tile] = total;          // the true count: a tile above slot_cap is redone into a spill region
                if (total > slot_cap) {
                    const uint32_t s = atomicAdd(&spill->n_tiles, 1u);
                    if (s < SPILL_MAX_TILES) spill->tiles[s] = tile;
                }
            }
        }
        __syncthreads();   // s_wave is rewritten by the next tile
    }
}

// out[tile_off[t] + i] = slot[t * slot_cap + i]: the survivors of the whole batch in ascending position order; tiles that
// overflowed their slots are taken from their spill region instead
__global__ __launch_bounds__(64) void compact_slots_kernel(const uint64_t* __restrict__ slot_hash, const uint32_t* __restrict__ slot_pos,
                                                           const uint32_t* __restrict__ tile_count,
                                                           const uint32_t* __restrict__ tile_off, uint32_t n_tiles, uint32_t slot_cap,
                                                           const uint64_t* __restrict__ spill_hash, const uint32_t* __restrict__ spill_pos,
                                                           const uint32_t* __restrict__ spill_slot_of_tile,
                                                           uint32_t* __restrict__ out_pos, uint64_t* __restrict__ out_hash) {
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t n = tile_count[t], d = tile_off[t];
        const uint64_t* sh = slot_hash + (uint64_t)t * slot_cap;
        const uint32_t* sp = slot_pos + (uint64_t)t * slot_cap;
        if (n > slot_cap) {
            const uint64_t s = (uint64_t)spill_slot_of_tile[t] * TILE_BASES;
            sh = spill_hash + s;
            sp = spill_pos + s;
        }
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) { out_pos[d + i] = sp[i]; out_hash[d + i] = sh[i]; }
    }
}

}  // namespace

// Launch K1 on ctx->stream.  d_count must be zeroed by the caller.  Returns nothing; caller reads *d_count.
void launch_seeds(sylph_ctx* ctx, const uint8_t* d_bases, uint32_t n_bases, uint32_t c, uint32_t k, uint64_t* d_out_hash,
                  uint32_t* d_out_pos, uint32_t out_cap, uint32_t* d_count) {
    if (n_bases == 0) return;
    const uint64_t thr = UINT64_MAX / (uint64_t)c;
    const uint32_t n_tiles = (uint32_t)(((uint64_t)n_bases + TILE_BASES - 1) / TILE_BASES);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)cus * 8);   // looping workgroups: the staging area is flushed with one global atomic per flush
    static const int hv = getenv("SYLPH_HIP_HASH_VARIANT") ? atoi(getenv("SYLPH_HIP_HASH_VARIANT")) : 1;   // tuning knob
    ctx->seed_gate();
    ScopedKernelTimer t(ctx, "seeds");
#define SY_LAUNCH_SEEDS(KK, HH)                                                                                         \
    hipLaunchKernelGGL((seeds_kernel<KK, HH>), dim3(grid), dim3(TPB), 0, ctx->stream, d_bases, n_bases, thr, n_tiles, \
                       d_out_hash, d_out_pos, out_cap, d_count)
    if (k == 31) { if (hv) SY_LAUNCH_SEEDS(31, 1); else SY_LAUNCH_SEEDS(31, 0); }
    else if (k == 21) { if (hv) SY_LAUNCH_SEEDS(21, 1); else SY_LAUNCH_SEEDS(21, 0); }
    else throw ArgError{"k must be 21 or 31 (avx2_seeding.rs:46-52)"};
#undef SY_LAUNCH_SEEDS
    SY_HIP(hipGetLastError());
    ctx->seed_done();
}


// Ordered K1: fills scratch slots + tile counts; the caller scans tile_count and calls launch_compact_slots.
uint32_t seeds_slot_capacity(uint32_t c) {
    const uint64_t expect = (uint64_t)TILE_BASES / c;
    return (uint32_t)std::min<uint64_t>(TILE_BASES, expect + expect * 3 / 4 + 48);
}
uint32_t seeds_n_tiles(uint64_t n_bases) { return (uint32_t)((n_bases + TILE_BASES - 1) / TILE_BASES); }

// tile_list == nullptr: first pass over all tiles of the batch.  Otherwise: redo pass over `n_list` overflowed tiles with
// TILE_BASES slots each (d_slot_* = the spill regions), recording each tile's spill slot in d_spill_slot_of_tile.
void launch_seeds_slots(sylph_ctx* ctx, const uint8_t* d_bases, uint32_t n_bases, uint32_t c, uint32_t k, uint32_t slot_cap,
                        uint64_t* d_slot_hash, uint32_t* d_slot_pos, uint32_t* d_tile_count, SpillState* d_spill,
                        const uint32_t* d_tile_list, uint32_t n_list, uint32_t* d_spill_slot_of_tile) {
    const uint64_t thr = UINT64_MAX / (uint64_t)c;
    const uint32_t n_tiles = d_tile_list ? n_list : seeds_n_tiles(n_bases);
    if (d_tile_list) slot_cap = TILE_BASES;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    const uint32_t grid = ctx->reads_wg_per_cu ? (uint32_t)std::min<uint64_t>(n_tiles, (uint64_t)cus * ctx->reads_wg_per_cu) : n_tiles;
    static const int hv = getenv("SYLPH_HIP_HASH_VARIANT") ? atoi(getenv("SYLPH_HIP_HASH_VARIANT")) : 1;   // tuning knob
    ctx->seed_gate();
    ScopedKernelTimer t(ctx, "seeds");
#define SY_LAUNCH_SLOTS(KK, HH)                                                                                              \
    hipLaunchKernelGGL((seeds_slots_kernel<KK, HH>), dim3(grid), dim3(TPB), 0, ctx->stream, d_bases, n_bases, thr, n_tiles, \
                       slot_cap, d_slot_hash, d_slot_pos, d_tile_count, d_spill, d_tile_list, d_spill_slot_of_tile)
    if (k == 31) { if (hv) SY_LAUNCH_SLOTS(31, 1); else SY_LAUNCH_SLOTS(31, 0); }
    else if (k == 21) { if (hv) SY_LAUNCH_SLOTS(21, 1); else SY_LAUNCH_SLOTS(21, 0); }
    else throw ArgError{"k must be 21 or 31 (avx2_seeding.rs:46-52)"};
#undef SY_LAUNCH_SLOTS
    SY_HIP(hipGetLastError());
    if (!d_tile_list) ctx->seed_done();
}
uint32_t seeds_tile_bases() { return TILE_BASES; }

void launch_compact_slots(sylph_ctx* ctx, const uint64_t* d_slot_hash, const uint32_t* d_slot_pos, const uint32_t* d_tile_count,
                          const uint32_t* d_tile_off, uint32_t n_tiles, uint32_t slot_cap, const uint64_t* d_spill_hash,
                          const uint32_t* d_spill_pos, const uint32_t* d_spill_slot_of_tile, uint32_t* d_out_pos,
                          uint64_t* d_out_hash) {
    ScopedKernelTimer t(ctx, "compact");
    hipLaunchKernelGGL(compact_slots_kernel, dim3(std::min<uint32_t>(n_tiles, 1u << 16)), dim3(64), 0, ctx->stream, d_slot_hash,
                       d_slot_pos, d_tile_count, d_tile_off, n_tiles, slot_cap, d_spill_hash, d_spill_pos, d_spill_slot_of_tile,
                       d_out_pos, d_out_hash);
    SY_HIP(hipGetLastError());
}

}  // namespace sylph
