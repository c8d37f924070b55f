// Stable LSD radix sort for gfx950 (wave64) — replaces resources/shaders/compute/radix_sort_{upsweep,spine,downsweep}.glsl.
//
// The reference's sort only works with 32-wide subgroups (radix_sort_spine.glsl:33,56-59,
// radix_sort_downsweep.glsl:144-156); what is kept is its CONTRACT: after the passes the (key, value) pairs are
// ascending by key and equal keys keep their emission order (radix_sort_downsweep.glsl:178-213:
// dst = global[digit] + partition[digit] + local rank).
//
// Keys are (tile << 16 | depth16) and every pair of a splat carries the splat's depth16, so the two low passes of the
// reference's four sort *splats*, not pairs: ordering the V visible splats by (depth16, id) and emitting their pairs in
// that order yields exactly the array the reference holds after its second pass.  The sort is therefore split:
//   splat level  launch_sort_splats: two 8-bit passes over V 12-byte elements {depth16 | origin tile << 16, id,
//                tile-rectangle size}; pass 0 reads the projection kernel's per-splat hand-off directly (its
//                per-workgroup digit histograms are computed by that kernel) and compacts away culled splats;
//   pair level   launch_sort_pairs: only the tile bits [16, 16 + ceil(log2 T)) — two passes up to 65 536 tiles.
// At D/N = 1.65 this moves 40 % fewer bytes than four pair passes, at D/N = 9.4 (a real capture's density) 50 % fewer.
//
// Mechanics of one pass (reduce-then-scan, native wave64):
//   upsweep   : per partition, 256-bin digit histogram in LDS (uint4 key loads)
//   spine     : one workgroup per digit, exclusive scan over partitions (+ digit totals)
//   downsweep : wave-striped key loads, match-any ranking with 8 x 64-bit ballots per key, per-wave
//               digit counters in LDS, workgroup scan, reorder through LDS, coalesced scatter in digit runs.
// Element counts live in device memory; grids are fixed and partitions are grid-strided, so there is no host
// read-back and no indirect dispatch (gaussian_splatting_rasterizer.gd:146-148 used dispatch_indirect for that).
#include "gsplat_internal.h"
#include "../../include/gsplat.h"

namespace gsplat {

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_BLOCK = 256;                 // 4 wave64
constexpr int SORT_WAVES = SORT_BLOCK / 64;
#ifndef GSPLAT_SORT_KPT
#define GSPLAT_SORT_KPT 16
#endif
constexpr int KPT = GSPLAT_SORT_KPT;            // pair passes: keys per lane -> 4096-key partitions
constexpr int KPT_SPLAT = SPLAT_PART0 / SORT_BLOCK;  // splat passes: 8 -> 2048-element partitions
constexpr int SORT_GRID = 2048;                 // 256 CUs x 8 workgroups
// Small inputs (a stripe of an 8-GPU shard, a 100 k-splat scene) are latency-bound: up to SMALL_COUNT elements the
// same kernels cut the input into 1024-element partitions (4 per lane): more workgroups, each shorter.  The choice
// is made on the device from the element count, identically in the three kernels of a pass; same result either way.
constexpr int KPT_SMALL = 4;
constexpr uint32_t SMALL_COUNT = 5u << 18;  // 1.3 M: measured crossover (tools/sort_small_sweep.py)

__device__ __host__ __forceinline__ uint32_t partitions_of(uint32_t count, uint32_t small_count, uint32_t part_big) {
    const uint32_t p = count <= small_count ? (uint32_t)(SORT_BLOCK * KPT_SMALL) : part_big;
    return (count + p - 1) / p;
}

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift, uint32_t mask = RADIX - 1) {
    return (key >> shift) & mask;
}

// one key into a workgroup histogram in LDS.  A wave whose 64 keys share the digit (sorted-ish input: the high tile
// bits of pairs that arrive grouped by the low ones) adds 64 with one atomic instead of a 64-way same-address conflict.
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t d) {
    const uint32_t first = __builtin_amdgcn_readfirstlane(d);
    if (__all(d == first)) {
        if ((threadIdx.x & 63) == 0) atomicAdd(&hist[first], 64u);
    } else {
        atomicAdd(&hist[d], 1u);
    }
}

// Which partitions a workgroup takes.  The dispatcher places workgroup b on XCD b % 8 (MI355X_MICROARCH.md; observed, only
// speed depends on it), and every XCD has its own L2.  A downsweep partition writes one short run per digit, and
// the runs of NEIGHBOURING partitions are adjacent in memory: XCD x therefore takes the contiguous eighth
// [x * per_xcd, (x + 1) * per_xcd) of the partitions, in ascending order over its workgroups, so the fragments of a
// 128-byte line meet in one L2 instead of leaving eight L2s as partial-line write-backs.
#ifndef GSPLAT_SORT_XCD_ORDER
#define GSPLAT_SORT_XCD_ORDER 1
#endif
struct PartitionWalk {
    uint32_t first, step, per_xcd, base, end;
    __device__ __forceinline__ PartitionWalk(uint32_t num_parts) {
#if GSPLAT_SORT_XCD_ORDER
        const uint32_t groups = gridDim.x >> 3;  // workgroups per XCD (the grids are multiples of 8 — or smaller than 8)
        if (groups == 0u) { first = blockIdx.x; step = gridDim.x; base = 0; per_xcd = num_parts; end = num_parts; return; }
        per_xcd = (num_parts + 7u) >> 3;
        base = (blockIdx.x & 7u) * per_xcd;
        end = min(base + per_xcd, num_parts);
        first = blockIdx.x >> 3;
        step = groups;
        if (blockIdx.x >= (groups << 3)) first = per_xcd;  // the grid's remainder above a multiple of 8 idles
#else
        first = blockIdx.x; step = gridDim.x; base = 0; per_xcd = num_parts; end = num_parts;
#endif
    }
};
#define GSPLAT_FOR_PARTITIONS(P, NUM)                 \
    const PartitionWalk walk_(NUM);                   \
    for (uint32_t q_ = walk_.first, P = walk_.base + q_; q_ < walk_.per_xcd && P < walk_.end; q_ += walk_.step, P = walk_.base + q_)

// part_hist is digit-major, part_hist[digit * stride + partition]: the spine scans contiguous rows.
constexpr int UPSWEEP_COPIES = 2;  // sub-histograms: spread the same-address LDS atomics of hot digits
template <int K>
__device__ __forceinline__ void upsweep_partitions(const uint32_t *__restrict__ keys, uint32_t count, int shift,
                                                   uint32_t mask, uint32_t *__restrict__ part_hist, uint32_t stride,
                                                   uint32_t (*hist)[RADIX]) {
    constexpr uint32_t P = SORT_BLOCK * K;
    const uint32_t num_parts = (count + P - 1) / P;
    uint32_t *my = hist[threadIdx.x & (UPSWEEP_COPIES - 1)];
    GSPLAT_FOR_PARTITIONS(p, num_parts) {
#pragma unroll
        for (int c = 0; c < UPSWEEP_COPIES; ++c) hist[c][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t start = p * P;
        if (start + P <= count) {
            const uint4 *src = reinterpret_cast<const uint4 *>(keys + start);
#pragma unroll
            for (int i = 0; i < K / 4; ++i) {
                const uint4 k = src[i * SORT_BLOCK + threadIdx.x];
                hist_add(my, digit_of(k.x, shift, mask));
                hist_add(my, digit_of(k.y, shift, mask));
                hist_add(my, digit_of(k.z, shift, mask));
                hist_add(my, digit_of(k.w, shift, mask));
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const uint32_t idx = start + i * SORT_BLOCK + threadIdx.x;
                if (idx < count) atomicAdd(&my[digit_of(keys[idx], shift, mask)], 1u);
            }
        }
        __syncthreads();
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < UPSWEEP_COPIES; ++c) v += hist[c][threadIdx.x];
        if (threadIdx.x <= mask) part_hist[(size_t)threadIdx.x * stride + p] = v;  // rows above the pass's digit range stay untouched
        __syncthreads();
    }
}

template <int KBIG>
__global__ __launch_bounds__(SORT_BLOCK) void upsweep_kernel(const uint32_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ d_count, int shift,
                                                             uint32_t mask, uint32_t *__restrict__ part_hist,
                                                             uint32_t stride, uint32_t small_count) {
    __shared__ uint32_t hist[UPSWEEP_COPIES][RADIX];
    const uint32_t count = *d_count;
    if (count <= small_count) upsweep_partitions<KPT_SMALL>(keys, count, shift, mask, part_hist, stride, hist);
    else upsweep_partitions<KBIG>(keys, count, shift, mask, part_hist, stride, hist);
}

// workgroup-wide exclusive scan of one u32 per lane (256 lanes); returns exclusive prefix, *total = sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot /*[SORT_WAVES]*/,
                                                         uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// One 1024-lane workgroup per digit: in-place exclusive scan of part_hist[digit][.] over partitions and
// digit_total[digit].  Each lane takes SPINE_ITEMS consecutive partitions per trip (4096 partitions per trip).
// d_count == nullptr: the partition count comes from the host (splat pass 0: one partition per projection workgroup).
constexpr int SPINE_BLOCK = 1024;
constexpr int SPINE_ITEMS = 4;
__global__ __launch_bounds__(SPINE_BLOCK) void spine_kernel(uint32_t *__restrict__ part_hist_all,
                                                            const uint32_t *__restrict__ d_count, uint32_t host_parts,
                                                            uint32_t *__restrict__ digit_total, uint32_t stride,
                                                            uint32_t small_count, uint32_t part_big) {
    __shared__ uint32_t wave_tot[SPINE_BLOCK / 64];
    const uint32_t num_parts = d_count ? partitions_of(*d_count, small_count, part_big) : host_parts;
    const uint32_t digit = blockIdx.x;
    uint32_t *part_hist = part_hist_all + (size_t)digit * stride;  // this digit's row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_parts; base += SPINE_BLOCK * SPINE_ITEMS) {
        const uint32_t p0 = base + threadIdx.x * SPINE_ITEMS;
        uint32_t v[SPINE_ITEMS], mine = 0;
#pragma unroll
        for (int k = 0; k < SPINE_ITEMS; ++k) {
            v[k] = (p0 + k) < num_parts ? part_hist[p0 + k] : 0u;
            mine += v[k];
        }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < SPINE_BLOCK / 64; ++w) {
            const uint32_t t = wave_tot[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        __syncthreads();
        uint32_t run = carry + wbase + incl - mine;
#pragma unroll
        for (int k = 0; k < SPINE_ITEMS; ++k) {
            if ((p0 + k) < num_parts) part_hist[p0 + k] = run;
            run += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) digit_total[digit] = carry;
}

// One element = a key word + NP payload words, structure-of-arrays.
template <int NP>
struct SortIO {
    const uint32_t *key_in;
    const uint32_t *pay_in[NP];
    uint32_t *key_out;
    uint32_t *pay_out[NP];
};

// LDS of a downsweep workgroup, carved from one array: per-wave digit counters -> exclusive wave prefixes,
// exclusive scan of the partition's digit counts, global base of each digit run minus that, the reorder buffers.
constexpr uint32_t DS_WAVE_CNT = 0, DS_LOCAL_START = SORT_WAVES * RADIX, DS_DST_BASE = DS_LOCAL_START + RADIX,
                   DS_WAVE_TOT = DS_DST_BASE + RADIX, DS_REORDER = DS_WAVE_TOT + 8;
__host__ __device__ constexpr uint32_t downsweep_lds_words(int k, int np) {
    return DS_REORDER + (uint32_t)(SORT_BLOCK * k) * (uint32_t)(1 + np);
}

// FIRST (splat pass 0): the input is the projection hand-off indexed by slot — payload 0 is the slot itself, payload 1
// the rectangle size, an element exists where that size is non-zero — and the per-partition histograms were written
// per 512-slot projection workgroup (hist_step of them per partition: the exclusive prefix of the first one applies).
template <int K, int NP, bool FIRST, int BITS>
__device__ __forceinline__ void downsweep_partitions(const SortIO<NP> &io, uint32_t count, int shift,
                                                     const uint32_t *__restrict__ part_hist, uint32_t stride,
                                                     uint32_t hist_step, uint32_t my_digit_base, uint32_t *smem) {
    constexpr uint32_t P = SORT_BLOCK * K;
    constexpr uint32_t WK = K * 64;  // elements per wave
    constexpr uint32_t MASK = (1u << BITS) - 1u;  // digits of BITS bits: BITS ballots per element
    uint32_t(*wave_cnt)[RADIX] = reinterpret_cast<uint32_t(*)[RADIX]>(smem + DS_WAVE_CNT);
    uint32_t *local_start = smem + DS_LOCAL_START, *dst_base = smem + DS_DST_BASE, *wave_tot = smem + DS_WAVE_TOT;
    uint32_t *lkeys = smem + DS_REORDER;
    const uint32_t num_parts = (count + P - 1) / P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    GSPLAT_FOR_PARTITIONS(p, num_parts) {
        const uint32_t start = p * P;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;

        uint32_t key[K], rank[K], first_dims[FIRST ? K : 1];
        bool ok[K];
        const uint32_t wbase = start + wave * WK + lane;
        const bool full = start + P <= count;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const uint32_t idx = wbase + r * 64;
            ok[r] = full || idx < count;
            key[r] = ok[r] ? io.key_in[idx] : 0u;  // (in range: loaded whether or not the slot holds an element)
            if constexpr (FIRST) {
                first_dims[r] = ok[r] ? io.pay_in[NP - 1][idx] : 0u;
                ok[r] = first_dims[r] != 0u;
            }
        }
        __syncthreads();  // counters zeroed

        // rank each element among this wave's earlier elements with the same digit (stable).  The counters are
        // re-read every round through a volatile pointer: other lanes of the wave update them.
        volatile uint32_t *my_cnt = wave_cnt[wave];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const uint32_t d = digit_of(key[r], shift, MASK);
            unsigned long long m = (FIRST || !full) ? __ballot(ok[r]) : ~0ull;
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            if (ok[r]) {
                const uint32_t before = my_cnt[d];
                const uint32_t in_group = (uint32_t)__popcll(m & lt_mask);
                const bool last = (m >> lane) <= 1ull;  // highest lane of the group
                rank[r] = before + in_group;
                if (last) my_cnt[d] = before + in_group + 1u;
            }
        }
        __syncthreads();

        // digit = threadIdx.x: wave-exclusive prefixes, partition digit count, scan over digits
        uint32_t valid;
        {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) {
                const uint32_t c = wave_cnt[w][threadIdx.x];
                wave_cnt[w][threadIdx.x] = run;
                run += c;
            }
            const uint32_t ls = block_exclusive_scan(run, wave_tot, &valid);
            local_start[threadIdx.x] = ls;
            const uint32_t before = threadIdx.x <= MASK ? part_hist[(size_t)threadIdx.x * stride + (size_t)p * hist_step] : 0u;
            dst_base[threadIdx.x] = my_digit_base + before - ls;
        }
        __syncthreads();

        // reorder through LDS so that each digit run leaves as contiguous, coalesced stores.  Payloads are loaded
        // only now (not before the ranking: fewer live registers through the ballot loops), all of them before the
        // first LDS write so that the loads are in flight together.
        uint32_t pay[NP][K];
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const uint32_t idx = wbase + r * 64;
                if constexpr (FIRST) pay[j][r] = j == 0 ? idx : first_dims[r];
                else pay[j][r] = ok[r] ? io.pay_in[j][idx] : 0u;
            }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (ok[r]) {
                const uint32_t d = digit_of(key[r], shift, MASK);
                const uint32_t pos = local_start[d] + wave_cnt[wave][d] + rank[r];
                lkeys[pos] = key[r];
#pragma unroll
                for (int j = 0; j < NP; ++j) lkeys[(uint32_t)(1 + j) * P + pos] = pay[j][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const uint32_t li = i * SORT_BLOCK + threadIdx.x;
            if (li < valid) {
                const uint32_t k = lkeys[li];
                const uint32_t dst = dst_base[digit_of(k, shift, MASK)] + li;
                io.key_out[dst] = k;
#pragma unroll
                for (int j = 0; j < NP; ++j) io.pay_out[j][dst] = lkeys[(uint32_t)(1 + j) * P + li];
            }
        }
        __syncthreads();
    }
}

// pair pass: (key, value), digits of BITS bits
template <int BITS>
__global__ __launch_bounds__(SORT_BLOCK) void downsweep_pairs_kernel(SortIO<1> io, const uint32_t *__restrict__ d_count,
                                                                     int shift, const uint32_t *__restrict__ part_hist,
                                                                     const uint32_t *__restrict__ digit_total,
                                                                     uint32_t stride, uint32_t small_count) {
    __shared__ uint32_t smem[downsweep_lds_words(KPT, 1)];
    const uint32_t count = *d_count;
    // exclusive scan of the pass's global digit histogram (identical in every workgroup)
    uint32_t unused;
    const uint32_t mine = threadIdx.x < (1u << BITS) ? digit_total[threadIdx.x] : 0u;
    const uint32_t my_digit_base = block_exclusive_scan(mine, smem + DS_WAVE_TOT, &unused);
    if (count <= small_count)
        downsweep_partitions<KPT_SMALL, 1, false, BITS>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
    else
        downsweep_partitions<KPT, 1, false, BITS>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
}

// splat passes: {depth16 | origin tile << 16, slot, rectangle size}
template <bool FIRST>
__global__ __launch_bounds__(SORT_BLOCK) void downsweep_splats_kernel(SortIO<2> io, const uint32_t *__restrict__ d_count,
                                                                      uint32_t host_count, int shift,
                                                                      const uint32_t *__restrict__ part_hist,
                                                                      const uint32_t *__restrict__ digit_total,
                                                                      uint32_t stride, uint32_t small_count,
                                                                      uint32_t *__restrict__ total_out) {
    __shared__ uint32_t smem[downsweep_lds_words(KPT_SPLAT, 2)];
    uint32_t total;
    const uint32_t my_digit_base = block_exclusive_scan(digit_total[threadIdx.x], smem + DS_WAVE_TOT, &total);
    if (FIRST) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = total;  // V: the splats that emit pairs this frame
        downsweep_partitions<KPT_SPLAT, 2, true, 8>(io, host_count, shift, part_hist, stride,
                                                    (uint32_t)(SPLAT_PART0 / PROJ_BLOCK), my_digit_base, smem);
    } else {
        const uint32_t count = *d_count;
        if (count <= small_count)
            downsweep_partitions<KPT_SMALL, 2, false, 8>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
        else
            downsweep_partitions<KPT_SPLAT, 2, false, 8>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
    }
}

// ---------------------------------------------------------------------------------------------------
// Key emission fused with the FIRST pair pass (gsplat_projection.glsl:216-226 + radix_sort_* pass 2 of the reference).
// The sorted splat list is a generator of the pairs in (depth16, id) order; writing them out in that order only to read
// them back for the pass on the low tile bits costs 8 + 4 + 8 bytes per pair.  Instead:
//   emit_hist_kernel     a workgroup takes EMIT_PART list entries, walks their tile rectangles and counts the digit
//                        (tile & mask) of every pair they will emit -> one histogram row per workgroup ("upsweep"
//                        without keys);
//   spine_kernel         as for any pass;
//   emit_scatter_kernel  the same workgroups generate their pairs chunk by chunk, y-outer / x-inner in list order, rank
//                        them by digit exactly like downsweep_partitions (ballots, per-wave counters, LDS reorder) and
//                        write them straight to where the pass would have put them.
// Same array as emit + pass, bit for bit; 20 of 48 bytes per pair of the pair-level sort are gone.  Pairs beyond the key
// budget are those with emission index >= capacity, as before: neither counted nor written.
// ---------------------------------------------------------------------------------------------------
#ifndef GSPLAT_EMIT_KPT
#define GSPLAT_EMIT_KPT 8
#endif
constexpr int EMIT_PART = 1024;       // list entries per emission workgroup = two blocks of emit_sums / block_base
constexpr uint32_t EMIT_BIG_TILES = 256;  // rectangles above this are counted by the whole workgroup, not by their lane

__device__ __forceinline__ uint32_t div_by(uint32_t j, uint32_t w, uint32_t &rem_out) {
    // j / w without an integer divide: float estimate (j < 2^24), corrected by at most one
    uint32_t q = (uint32_t)((float)j * (1.0f / (float)w));
    int32_t rem = (int32_t)(j - q * w);
    if (rem < 0) { --q; rem += (int32_t)w; }
    if (rem >= (int32_t)w) { ++q; rem -= (int32_t)w; }
    rem_out = (uint32_t)rem;
    return q;
}

__global__ __launch_bounds__(SORT_BLOCK) void emit_hist_kernel(SplatList list, const uint32_t *__restrict__ v_count,
                                                               uint32_t gx, const uint32_t *__restrict__ emit_sums,
                                                               const uint64_t *__restrict__ block_base,
                                                               uint64_t capacity, uint32_t mask,
                                                               uint32_t *__restrict__ part_hist, uint32_t stride) {
    __shared__ uint32_t hist[RADIX];
    __shared__ uint32_t wave_tot[SORT_WAVES];
    __shared__ uint32_t big_n, big_entry[EMIT_PART], big_keep[EMIT_PART];
    const uint32_t v = *v_count;
    const uint32_t num_parts = (v + EMIT_PART - 1) / EMIT_PART;
    GSPLAT_FOR_PARTITIONS(part, num_parts) {
        const uint32_t first = part * EMIT_PART;
        hist[threadIdx.x] = 0u;
        if (threadIdx.x == 0) big_n = 0u;
        __syncthreads();
        const uint64_t g0 = block_base[2u * part];
        const uint64_t total = (uint64_t)emit_sums[2u * part] + ((2u * part + 1u) * 512u < v ? emit_sums[2u * part + 1u] : 0u);
        const bool guard = g0 + total > capacity;  // workgroup-uniform: the key budget ends inside this partition
        uint32_t carry = 0;
#pragma unroll 1
        for (uint32_t r = 0; r < EMIT_PART / SORT_BLOCK; ++r) {
            const uint32_t e = first + r * SORT_BLOCK + threadIdx.x;
            uint32_t w = 1, h = 0, t0 = 0;
            if (e < v) {
                const uint32_t d = list.dims[e];
                w = d & 0xFFFFu; h = d >> 16;
                t0 = list.key[e] >> 16;
            }
            const uint32_t count = w * h;
            uint32_t keep = count;
            if (guard) {
                uint32_t tot;
                const uint32_t excl = block_exclusive_scan(count, wave_tot, &tot);
                const uint64_t off = g0 + carry + excl;
                keep = off >= capacity ? 0u : (uint32_t)min((uint64_t)count, capacity - off);
                carry += tot;
            }
            if (keep > EMIT_BIG_TILES) {
                const uint32_t slot = atomicAdd(&big_n, 1u);
                big_entry[slot] = e;
                big_keep[slot] = keep;
            } else {
                uint32_t j = 0;
                for (uint32_t row = 0; j < keep; ++row) {
                    const uint32_t rb = t0 + row * gx;
                    for (uint32_t c = 0; c < w && j < keep; ++c, ++j) atomicAdd(&hist[(rb + c) & mask], 1u);
                }
            }
        }
        __syncthreads();
        const uint32_t nb = big_n;
        for (uint32_t b = 0; b < nb; ++b) {  // big rectangles: the workgroup strides over the tiles
            const uint32_t e = big_entry[b], keep = big_keep[b];
            const uint32_t w = list.dims[e] & 0xFFFFu, t0 = list.key[e] >> 16;
            for (uint32_t j = threadIdx.x; j < keep; j += SORT_BLOCK) {
                uint32_t rem;
                const uint32_t q = div_by(j, w, rem);
                atomicAdd(&hist[(t0 + q * gx + rem) & mask], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x <= mask) part_hist[(size_t)threadIdx.x * stride + part] = hist[threadIdx.x];
        __syncthreads();
    }
}

constexpr uint32_t ES_RUNNING = DS_REORDER, ES_ENT = ES_RUNNING + RADIX, ES_REORDER = ES_ENT + EMIT_PART * 4;
__host__ __device__ constexpr uint32_t emit_scatter_lds_words(int k) { return ES_REORDER + 2u * (uint32_t)(SORT_BLOCK * k); }

template <int BITS, int K>
__global__ __launch_bounds__(SORT_BLOCK) void emit_scatter_kernel(SplatList list, const uint32_t *__restrict__ v_count,
                                                                  uint32_t gx, const uint32_t *__restrict__ emit_sums,
                                                                  const uint64_t *__restrict__ block_base,
                                                                  uint64_t capacity,
                                                                  const uint32_t *__restrict__ part_hist,
                                                                  const uint32_t *__restrict__ digit_total,
                                                                  uint32_t stride, uint32_t *__restrict__ keys_out,
                                                                  uint32_t *__restrict__ vals_out) {
    constexpr uint32_t P = SORT_BLOCK * K, WK = K * 64, MASK = (1u << BITS) - 1u;
    __shared__ uint32_t smem[emit_scatter_lds_words(K)];
    __shared__ uint32_t s_before;
    uint32_t(*wave_cnt)[RADIX] = reinterpret_cast<uint32_t(*)[RADIX]>(smem + DS_WAVE_CNT);
    uint32_t *local_start = smem + DS_LOCAL_START, *dst_base = smem + DS_DST_BASE, *wave_tot = smem + DS_WAVE_TOT;
    uint32_t *running = smem + ES_RUNNING;                           // pairs of each digit in the partition's earlier chunks
    uint4 *ent = reinterpret_cast<uint4 *>(smem + ES_ENT);           // {key word, slot, dims, first pair index} per entry
    uint32_t *lkeys = smem + ES_REORDER, *lvals = lkeys + P;         // reorder buffers ...
    uint32_t *owner = lkeys;                                         // ... and, before them in a chunk, pair -> entry
    const uint32_t v = *v_count;
    const uint32_t num_parts = (v + EMIT_PART - 1) / EMIT_PART;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t unused;
    const uint32_t my_digit_base =
        block_exclusive_scan(threadIdx.x <= MASK ? digit_total[threadIdx.x] : 0u, wave_tot, &unused);
    GSPLAT_FOR_PARTITIONS(part, num_parts) {
        const uint32_t first = part * EMIT_PART;
        const uint32_t nent = min((uint32_t)EMIT_PART, v - first);
        // the partition's entries and the index of each one's first pair (4 consecutive entries per lane)
        uint32_t ekey[4], eid[4], edim[4], ecnt[4], mine = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t el = threadIdx.x * 4u + k;
            ekey[k] = eid[k] = edim[k] = ecnt[k] = 0u;
            if (el < nent) {
                ekey[k] = list.key[first + el];
                eid[k] = list.id[first + el];
                edim[k] = list.dims[first + el];
                ecnt[k] = (edim[k] & 0xFFFFu) * (edim[k] >> 16);
            }
            mine += ecnt[k];
        }
        uint32_t total;
        uint32_t pre = block_exclusive_scan(mine, wave_tot, &total);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ent[threadIdx.x * 4u + k] = make_uint4(ekey[k], eid[k], edim[k], pre);
            pre += ecnt[k];
        }
        running[threadIdx.x] = 0u;
        const uint64_t g0 = block_base[2u * part];
        const uint32_t limit = g0 + total <= capacity ? total : (g0 >= capacity ? 0u : (uint32_t)(capacity - g0));
        const uint32_t part_base = threadIdx.x <= MASK ? my_digit_base + part_hist[(size_t)threadIdx.x * stride + part] : 0u;
        __syncthreads();

        for (uint32_t c0 = 0; c0 < limit; c0 += P) {
            // which entry owns pair c0 + i?  mark the first pair of every entry that starts inside the chunk, count the
            // entries that start before it, inclusive-scan the marks
            for (uint32_t i = threadIdx.x; i < P; i += SORT_BLOCK) owner[i] = 0u;
            if (threadIdx.x == 0) {  // entries whose first pair lies before c0 (first pair indices ascend)
                uint32_t lo = 0, hi = nent;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ent[mid].w < c0) lo = mid + 1; else hi = mid;
                }
                s_before = lo;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t el = threadIdx.x * 4u + k;
                if (el < nent && ecnt[k]) {
                    const uint32_t pos = ent[el].w;
                    if (pos >= c0 && pos - c0 < P) owner[pos - c0] = 1u;
                }
            }
            __syncthreads();
            {
                uint32_t m[K], sum = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) { m[k] = owner[threadIdx.x * K + k]; sum += m[k]; }
                uint32_t tot;
                uint32_t run = block_exclusive_scan(sum, wave_tot, &tot) + s_before - 1u;
#pragma unroll
                for (int k = 0; k < K; ++k) { run += m[k]; owner[threadIdx.x * K + k] = run; }
            }
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
            __syncthreads();

            // generate this lane's K pairs (wave-striped: the ranking order is the pair order) ...
            uint32_t key[K], val[K], rank[K];
            bool ok[K];
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const uint32_t pl = wave * WK + r * 64 + lane, p = c0 + pl;
                ok[r] = p < limit;
                key[r] = val[r] = 0u;
                if (ok[r]) {
                    const uint4 en = ent[owner[pl]];
                    uint32_t rem;
                    const uint32_t q = div_by(p - en.w, en.z & 0xFFFFu, rem);
                    key[r] = ((((en.x >> 16) + q * gx + rem)) << 16) | (en.x & 0xFFFFu);  // :222 (tile << 16) | depth16
                    val[r] = en.y;
                }
            }
            // ... rank them by digit (stable), as downsweep_partitions does
            volatile uint32_t *my_cnt = wave_cnt[wave];
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const uint32_t d = digit_of(key[r], 16, MASK);
                unsigned long long m = __ballot(ok[r]);
#pragma unroll
                for (int b = 0; b < BITS; ++b) {
                    const bool bit = (d >> b) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                if (ok[r]) {
                    const uint32_t before = my_cnt[d];
                    const uint32_t in_group = (uint32_t)__popcll(m & lt_mask);
                    rank[r] = before + in_group;
                    if ((m >> lane) <= 1ull) my_cnt[d] = before + in_group + 1u;
                }
            }
            __syncthreads();  // (also: every owner[] read is done before the reorder buffers overwrite it)
            uint32_t valid;
            {
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < SORT_WAVES; ++w) {
                    const uint32_t c = wave_cnt[w][threadIdx.x];
                    wave_cnt[w][threadIdx.x] = run;
                    run += c;
                }
                const uint32_t ls = block_exclusive_scan(run, wave_tot, &valid);
                local_start[threadIdx.x] = ls;
                const uint32_t before = running[threadIdx.x];
                dst_base[threadIdx.x] = part_base + before - ls;
                running[threadIdx.x] = before + run;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < K; ++r) {
                if (ok[r]) {
                    const uint32_t d = digit_of(key[r], 16, MASK);
                    const uint32_t pos = local_start[d] + wave_cnt[wave][d] + rank[r];
                    lkeys[pos] = key[r];
                    lvals[pos] = val[r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const uint32_t li = i * SORT_BLOCK + threadIdx.x;
                if (li < valid) {
                    const uint32_t k = lkeys[li];
                    const uint32_t dst = dst_base[digit_of(k, 16, MASK)] + li;
                    keys_out[dst] = k;
                    vals_out[dst] = lvals[li];
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

uint32_t grid_for(uint64_t max_parts) {  // a multiple of 8 (one share per XCD) once there are 8 partitions
    const uint32_t g = max_parts < (uint64_t)SORT_GRID ? (uint32_t)(max_parts ? max_parts : 1u) : (uint32_t)SORT_GRID;
    return g < 8u ? g : ((g + 7u) & ~7u);
}

}  // namespace

int sort_num_passes(int sig_bits) {
    if (sig_bits < 1) sig_bits = 1;
    if (sig_bits > 32) sig_bits = 32;
    return (sig_bits + RADIX_BITS - 1) / RADIX_BITS;
}

uint32_t sort_small_count_default() { return SMALL_COUNT; }

uint32_t sort_max_partitions(uint64_t capacity) {
    // the larger of: every pair in 4096-key partitions; as many pairs as the small mode takes, in 1024-key partitions
    const uint64_t part = (uint64_t)SORT_BLOCK * KPT, part_small = (uint64_t)SORT_BLOCK * KPT_SMALL;
    const uint64_t big = (capacity + part - 1) / part;
    const uint64_t small_pairs = capacity < SMALL_COUNT ? capacity : SMALL_COUNT;
    const uint64_t small = (small_pairs + part_small - 1) / part_small;
    return (uint32_t)(big > small ? big : small);
}

void launch_sort_splats(SortBuffers &sb, const SplatKeys &keys, uint32_t n, hipStream_t s, KernelTimer *kt) {
    if (n == 0) {
        (void)hipMemsetAsync(sb.v_count, 0, sizeof(uint32_t), s);
        return;
    }
    const uint32_t stride = (n + PROJ_BLOCK - 1) / PROJ_BLOCK;  // row length of splat_hist: one entry per projection workgroup
    const uint32_t small = sb.small_count;
    // pass 0 (depth16 & 255): histograms by the projection kernel; compaction of the visible splats
    hipLaunchKernelGGL(spine_kernel, dim3(RADIX), dim3(SPINE_BLOCK), 0, s, sb.splat_hist,
                       static_cast<const uint32_t *>(nullptr), stride, sb.digit_base, stride, 0u, (uint32_t)SPLAT_PART0);
    SortIO<2> io0{};
    io0.key_in = keys.key; io0.pay_in[0] = nullptr; io0.pay_in[1] = keys.dims;
    io0.key_out = sb.list[1].key; io0.pay_out[0] = sb.list[1].id; io0.pay_out[1] = sb.list[1].dims;
    const uint32_t parts0 = (n + SPLAT_PART0 - 1) / SPLAT_PART0;
    hipLaunchKernelGGL(downsweep_splats_kernel<true>, dim3(grid_for(parts0)), dim3(SORT_BLOCK), 0, s, io0,
                       static_cast<const uint32_t *>(nullptr), n, 0, sb.splat_hist, sb.digit_base, stride, 0u,
                       sb.v_count);
    // pass 1 (depth16 >> 8) over the compact list
    const uint32_t parts1 = (n + SORT_BLOCK * KPT_SMALL - 1) / (SORT_BLOCK * KPT_SMALL);
    hipLaunchKernelGGL(upsweep_kernel<KPT_SPLAT>, dim3(grid_for(parts1)), dim3(SORT_BLOCK), 0, s, sb.list[1].key,
                       sb.v_count, 8, (uint32_t)(RADIX - 1), sb.splat_hist, stride, small);
    hipLaunchKernelGGL(spine_kernel, dim3(RADIX), dim3(SPINE_BLOCK), 0, s, sb.splat_hist, sb.v_count, 0u,
                       sb.digit_base, stride, small, (uint32_t)SPLAT_PART0);
    SortIO<2> io1{};
    io1.key_in = sb.list[1].key; io1.pay_in[0] = sb.list[1].id; io1.pay_in[1] = sb.list[1].dims;
    io1.key_out = sb.list[0].key; io1.pay_out[0] = sb.list[0].id; io1.pay_out[1] = sb.list[0].dims;
    hipLaunchKernelGGL(downsweep_splats_kernel<false>, dim3(grid_for(parts1)), dim3(SORT_BLOCK), 0, s, io1,
                       sb.v_count, 0u, 8, sb.splat_hist, sb.digit_base, stride, small,
                       static_cast<uint32_t *>(nullptr));
    if (kt) kt->mark(GSPLAT_KERNEL_SPLAT_SORT);
}

int emit_first_pass_bits(int sig_bits) {
    // the tile bits [16, sig_bits) in the fewest passes of at most 8 bits, spread evenly; this is the first one's width
    const int total = sig_bits > 16 ? sig_bits - 16 : 1;
    const int passes = sort_num_passes(total);
    int bits = (total + passes - 1) / passes;
    return bits < 4 ? 4 : bits;  // (key bits above sig_bits are zero: a wider digit is the same digit)
}

void launch_emit_sorted(SortBuffers &sb, const uint32_t *v_count, uint32_t n, uint32_t gx, const uint32_t *emit_sums,
                        const uint64_t *block_base, uint64_t capacity, int sig_bits, hipStream_t s, KernelTimer *kt) {
    if (n == 0) return;
    const int bits = emit_first_pass_bits(sig_bits);
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t parts = (n + EMIT_PART - 1) / EMIT_PART;
    const uint32_t grid = grid_for(parts);
    hipLaunchKernelGGL(emit_hist_kernel, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.list[0], v_count, gx, emit_sums, block_base,
                       capacity, mask, sb.part_hist, sb.part_stride);
    hipLaunchKernelGGL(spine_kernel, dim3(mask + 1u), dim3(SPINE_BLOCK), 0, s, sb.part_hist, v_count, 0u, sb.digit_base,
                       sb.part_stride, 0u, (uint32_t)EMIT_PART);
    if (kt) kt->mark(GSPLAT_KERNEL_SCAN);
#define GSPLAT_LAUNCH_E(B)                                                                                              \
    hipLaunchKernelGGL((emit_scatter_kernel<B, GSPLAT_EMIT_KPT>), dim3(grid), dim3(SORT_BLOCK), 0, s, sb.list[0], v_count, \
                       gx, emit_sums, block_base, capacity, sb.part_hist, sb.digit_base, sb.part_stride, sb.keys[0],    \
                       sb.values[0])
    switch (bits) {
        case 4: GSPLAT_LAUNCH_E(4); break;
        case 5: GSPLAT_LAUNCH_E(5); break;
        case 6: GSPLAT_LAUNCH_E(6); break;
        case 7: GSPLAT_LAUNCH_E(7); break;
        default: GSPLAT_LAUNCH_E(8); break;
    }
#undef GSPLAT_LAUNCH_E
    if (kt) kt->mark(GSPLAT_KERNEL_EMIT);
}

int launch_sort_pairs(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int sig_bits, hipStream_t s,
                      KernelTimer *kt, int first_bit) {
    // the bits [first_bit, sig_bits) in the fewest passes of at most 8 bits, spread evenly (13 tile bits = 7 + 6:
    // fewer ballots per key and longer digit runs than 8 + 5)
    const int total = sig_bits > first_bit ? sig_bits - first_bit : 0;
    const int passes = total ? sort_num_passes(total) : 0;
    const uint32_t max_parts = sort_max_partitions(capacity);
    const uint32_t grid = grid_for(max_parts);
    const uint32_t stride = sb.part_stride;
    int cur = 0, shift = first_bit;
    for (int pass = 0; pass < passes; ++pass) {
        int bits = total / passes + (pass < total % passes ? 1 : 0);
        if (bits < 4) bits = 4;  // (key bits above sig_bits are zero: a wider digit is the same digit)
        if (shift + bits > 32) bits = 32 - shift;
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(upsweep_kernel<KPT>, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.keys[cur], d_count, shift, mask,
                           sb.part_hist, stride, sb.small_count);
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_UPSWEEP);
        hipLaunchKernelGGL(spine_kernel, dim3(mask + 1u), dim3(SPINE_BLOCK), 0, s, sb.part_hist, d_count, 0u,
                           sb.digit_base, stride, sb.small_count, (uint32_t)(SORT_BLOCK * KPT));
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_SPINE);
        SortIO<1> io{};
        io.key_in = sb.keys[cur]; io.pay_in[0] = sb.values[cur];
        io.key_out = sb.keys[cur ^ 1]; io.pay_out[0] = sb.values[cur ^ 1];
#define GSPLAT_LAUNCH_D(B)                                                                                          \
    hipLaunchKernelGGL(downsweep_pairs_kernel<B>, dim3(grid), dim3(SORT_BLOCK), 0, s, io, d_count, shift, sb.part_hist, \
                       sb.digit_base, stride, sb.small_count)
        switch (bits) {
            case 4: GSPLAT_LAUNCH_D(4); break;
            case 5: GSPLAT_LAUNCH_D(5); break;
            case 6: GSPLAT_LAUNCH_D(6); break;
            case 7: GSPLAT_LAUNCH_D(7); break;
            default: GSPLAT_LAUNCH_D(8); break;
        }
#undef GSPLAT_LAUNCH_D
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_DOWNSWEEP);
        shift += bits;
        cur ^= 1;
    }
    return cur;
}

}  // namespace gsplat
