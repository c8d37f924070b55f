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
//   pair level   only the tile bits are left to sort.  16-bit keys (the default) are STRIPE-LOCAL tile ids
//                (gsplat_internal.h: TileMap), 32-bit keys the reference's (tile << 16 | depth16);
//                launch_sort_pairs: ceil(bits / 8) stable passes of evenly spread width (13 bits = 7 + 6) —
//                launch_sort_pairs_wide: ONE counting-sort pass on the whole id for stripes of up to 4096 tiles and
//                modest pair counts (a stripe rank of a multi-GPU frame, a small frame): see "wide pass" below.
// At D/N = 1.65 this moves 40 % fewer bytes than four pair passes, at D/N = 9.4 (a real capture's density) 50 % fewer.
//
// Mechanics of one split pass (reduce-then-scan, native wave64):
//   upsweep   : per partition, digit histogram in LDS (16-byte key loads, wave-aggregated atomics, two sub-histograms)
//   spine     : one workgroup per digit, exclusive scan over partitions (+ digit totals)
//   downsweep : wave-striped key loads; the stable rank of a key among its wave's earlier keys with the same digit is ONE
//               returning LDS atomic on per-wave counters (lane-ordered on this hardware: self-tested per device, with
//               the match-any ballot form as fallback), workgroup scan, reorder through LDS, coalesced scatter in digit
//               runs; partitions walked XCD by XCD (PartitionWalk) so that neighbouring runs meet in one L2.
// Element counts live in device memory; grids are fixed and partitions are grid-strided, so there is no host
// read-back and no indirect dispatch (gaussian_splatting_rasterizer.gd:146-148 used dispatch_indirect for that).
#include <cstdlib>
#include <cstring>

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
// ... and 16 -> 4096-element partitions for scenes of SPLAT_BIG_N splats and more.  Pass 0's digit is the LOW depth byte,
// uniform over its 256 values: a 2048-element partition leaves runs of 8 elements = 32 B per plane, and at 30 M splats (c5)
// the XCD-contiguous walk no longer merges the neighbours' runs in the L2 before they are evicted — WRITE_SIZE 624 MB for 344
// MB of output.  Twice the partition: 64-byte runs, the two splat passes 0.533 -> 0.461 ms at c5 (365 -> 370 fps); at 6 M
// splats (c3) the smaller partitions win (0.094 vs 0.101 ms: occupancy, 29 vs 54 KiB of LDS per workgroup) —
// profiles/r05_ab_call4_splat_part4096.jsonl.
constexpr int KPT_SPLAT_BIG = 2 * KPT_SPLAT;
constexpr uint32_t SPLAT_BIG_N = 12u << 20;
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
template <int K, typename KeyT>
__device__ __forceinline__ void upsweep_partitions(const KeyT *__restrict__ keys, uint32_t count, int shift,
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
            // K keys per lane in 16-byte loads (8 bytes when a lane's share is only that: 4 narrow keys)
            constexpr int LANE_BYTES = K * (int)sizeof(KeyT), LOAD_BYTES = LANE_BYTES < 16 ? LANE_BYTES : 16;
            static_assert(LOAD_BYTES == 16 || LOAD_BYTES == 8, "a lane's keys are loaded 8 or 16 bytes at a time");
            static_assert(LANE_BYTES % LOAD_BYTES == 0, "keys per lane x key size must be a whole number of loads");
            constexpr int WORDS = LOAD_BYTES / 4;
#pragma unroll
            for (int i = 0; i < LANE_BYTES / LOAD_BYTES; ++i) {
                uint32_t w[WORDS];
                if constexpr (WORDS == 4) {
                    const uint4 k = reinterpret_cast<const uint4 *>(keys + start)[i * SORT_BLOCK + threadIdx.x];
                    w[0] = k.x; w[1] = k.y; w[2] = k.z; w[3] = k.w;
                } else {
                    const uint2 k = reinterpret_cast<const uint2 *>(keys + start)[i * SORT_BLOCK + threadIdx.x];
                    w[0] = k.x; w[1] = k.y;
                }
#pragma unroll
                for (int e = 0; e < WORDS; ++e) {
                    if constexpr (sizeof(KeyT) == 2) {
                        hist_add(my, digit_of(w[e] & 0xFFFFu, shift, mask));
                        hist_add(my, digit_of(w[e] >> 16, shift, mask));
                    } else {
                        hist_add(my, digit_of(w[e], shift, mask));
                    }
                }
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

template <int KBIG, typename KeyT>
__global__ __launch_bounds__(SORT_BLOCK) void upsweep_kernel(const KeyT *__restrict__ keys,
                                                             const uint32_t *__restrict__ d_count, int shift,
                                                             uint32_t mask, uint32_t *__restrict__ part_hist,
                                                             uint32_t stride, uint32_t small_count) {
    __shared__ uint32_t hist[UPSWEEP_COPIES][RADIX];
    const uint32_t count = *d_count;
    if (count <= small_count) upsweep_partitions<KPT_SMALL, KeyT>(keys, count, shift, mask, part_hist, stride, hist);
    else upsweep_partitions<KBIG, KeyT>(keys, count, shift, mask, part_hist, stride, hist);
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
// skip (nullable; splat pass 0 with block culling): partition p's column was not written this frame — it counts as zero.
__global__ __launch_bounds__(SPINE_BLOCK) void spine_kernel(uint32_t *__restrict__ part_hist_all,
                                                            const uint32_t *__restrict__ d_count, uint32_t host_parts,
                                                            uint32_t *__restrict__ digit_total, uint32_t stride,
                                                            uint32_t small_count, uint32_t part_big,
                                                            const uint32_t *__restrict__ skip) {
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
            v[k] = ((p0 + k) < num_parts && !(skip != nullptr && skip[p0 + k] != 0u)) ? part_hist[p0 + k] : 0u;
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

// One element = a key + NP payload words, structure-of-arrays.  KeyT = uint16_t: the pair passes of a frame whose
// tile ids fit 16 bits carry the tile id alone (the depth half of the reference's key no longer orders anything at
// the pair level: it did its work in the splat passes) — 6 instead of 8 bytes per pair read and written.
template <int NP, typename KeyT = uint32_t>
struct SortIO {
    const KeyT *key_in;
    const uint32_t *pay_in[NP];
    KeyT *key_out;
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
// skip (FIRST only, nullable): per 512-slot projection workgroup, 1 = culled this frame: its slots hold no element
// (whatever stale rectangle sizes they carry) — a partition all of whose workgroups were culled is left at once.
template <int K, int NP, bool FIRST, int BITS, bool ATOMIC_RANK, typename KeyT = uint32_t>
__device__ __forceinline__ void downsweep_partitions(const SortIO<NP, KeyT> &io, uint32_t count, int shift,
                                                     const uint32_t *__restrict__ part_hist, uint32_t stride,
                                                     uint32_t hist_step, uint32_t my_digit_base, uint32_t *smem,
                                                     const uint32_t *__restrict__ skip = nullptr,
                                                     const uint32_t *__restrict__ part_list = nullptr,
                                                     uint32_t part_list_count = 0) {
    constexpr uint32_t P = SORT_BLOCK * K;
    constexpr uint32_t WK = K * 64;  // elements per wave
    constexpr uint32_t MASK = (1u << BITS) - 1u;  // digits of BITS bits: BITS ballots per element
    uint32_t(*wave_cnt)[RADIX] = reinterpret_cast<uint32_t(*)[RADIX]>(smem + DS_WAVE_CNT);
    uint32_t *local_start = smem + DS_LOCAL_START, *dst_base = smem + DS_DST_BASE, *wave_tot = smem + DS_WAVE_TOT;
    uint32_t *lkeys = smem + DS_REORDER;
    const uint32_t num_parts = (count + P - 1) / P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // FIRST with a list (the frame's live partitions, projection.hip live_lists_kernel): the walk deals the eighths of the
    // LIST to the XCDs — a stripe rank skips two thirds of its partitions and the live ones cluster, so the eighths of the
    // index range were up to 3.4 x the mean
    const bool listed = FIRST && part_list != nullptr;
    GSPLAT_FOR_PARTITIONS(pw, (listed ? part_list_count : num_parts)) {
        const uint32_t p = listed ? part_list[pw] : pw;
        const uint32_t start = p * P;
        if constexpr (FIRST) {
            if (skip != nullptr && !listed) {  // (workgroup-uniform: every lane reads the same words)
                bool any = false;
                for (uint32_t b = start / PROJ_BLOCK; b < (start + P) / PROJ_BLOCK && b * PROJ_BLOCK < count; ++b) any = any || skip[b] == 0u;
                if (!any) continue;
            }
        }
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;

        uint32_t key[K], rank[K], first_dims[FIRST ? K : 1];
        bool ok[K];
        const uint32_t wbase = start + wave * WK + lane;
        const bool full = start + P <= count;
        // (this digit's exclusive prefix over the partitions before p: needed after the ranking, requested now)
        const uint32_t hist_before = threadIdx.x <= MASK ? part_hist[(size_t)threadIdx.x * stride + (size_t)p * hist_step] : 0u;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const uint32_t idx = wbase + r * 64;
            ok[r] = full || idx < count;
            key[r] = ok[r] ? (uint32_t)io.key_in[idx] : 0u;  // (in range: loaded whether or not the slot holds an element)
            if constexpr (FIRST) {
                // (a wave's 64 slots lie in one projection workgroup: the mark is the same for all its lanes)
                const bool live = ok[r] && !(skip != nullptr && skip[idx / PROJ_BLOCK] != 0u);
                first_dims[r] = live ? io.pay_in[NP - 1][idx] : 0u;
                ok[r] = first_dims[r] != 0u;
            }
        }
        // payloads requested with the keys: the ranking no longer holds BITS ballot masks live, and the LDS footprint
        // (not registers) sets the occupancy of these kernels
        uint32_t pay[NP][K];
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const uint32_t idx = wbase + r * 64;
                if constexpr (FIRST) pay[j][r] = j == 0 ? idx : first_dims[r];
                else pay[j][r] = (full || idx < count) ? io.pay_in[j][idx] : 0u;
            }
        __syncthreads();  // counters zeroed

        // rank each element among this wave's earlier elements with the same digit (stable)
        if constexpr (ATOMIC_RANK) {
        // One returning LDS atomic per element.  The LDS unit resolves the same-address lanes of ONE wave instruction
        // in ascending lane order and successive instructions of a wave in program order, so the value returned is
        // exactly "elements of this wave with my digit that come before me" (checked once per process against the
        // ballot form, sort_rank_selftest; a chip that resolves them differently falls back to that form).
#pragma unroll
        for (int r = 0; r < K; ++r)
            if (ok[r]) rank[r] = atomicAdd(&wave_cnt[wave][digit_of(key[r], shift, MASK)], 1u);
        } else {
        // BITS ballots per element (match-any).  The counters are re-read every round through a volatile pointer:
        // other lanes of the wave update them.
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
            dst_base[threadIdx.x] = my_digit_base + hist_before - ls;
        }
        __syncthreads();

        // reorder through LDS so that each digit run leaves as contiguous, coalesced stores
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
                io.key_out[dst] = (KeyT)k;
#pragma unroll
                for (int j = 0; j < NP; ++j) io.pay_out[j][dst] = lkeys[(uint32_t)(1 + j) * P + li];
            }
        }
        __syncthreads();
    }
}

// pair pass: (key, value), digits of BITS bits
template <int BITS, bool ATOMIC_RANK, typename KeyT>
__global__ __launch_bounds__(SORT_BLOCK) void downsweep_pairs_kernel(SortIO<1, KeyT> io, const uint32_t *__restrict__ d_count,
                                                                     int shift, const uint32_t *__restrict__ part_hist,
                                                                     const uint32_t *__restrict__ digit_total,
                                                                     uint32_t stride, uint32_t small_count) {
    __shared__ uint32_t smem[downsweep_lds_words(KPT, 1)];
    const uint32_t count = *d_count;
    if (count == 0u) return;  // (round B of a frame whose round A finished every tile: nothing to sort)
    // exclusive scan of the pass's global digit histogram (identical in every workgroup)
    uint32_t unused;
    const uint32_t mine = threadIdx.x < (1u << BITS) ? digit_total[threadIdx.x] : 0u;
    const uint32_t my_digit_base = block_exclusive_scan(mine, smem + DS_WAVE_TOT, &unused);
    if (count <= small_count)
        downsweep_partitions<KPT_SMALL, 1, false, BITS, ATOMIC_RANK, KeyT>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
    else
        downsweep_partitions<KPT, 1, false, BITS, ATOMIC_RANK, KeyT>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
}

// splat passes: {depth16 | origin tile << 16, slot, rectangle size}
template <bool FIRST, bool ATOMIC_RANK, int KS>
__global__ __launch_bounds__(SORT_BLOCK) void downsweep_splats_kernel(SortIO<2> io, const uint32_t *__restrict__ d_count,
                                                                      uint32_t host_count, int shift,
                                                                      const uint32_t *__restrict__ part_hist,
                                                                      const uint32_t *__restrict__ digit_total,
                                                                      uint32_t stride, uint32_t small_count,
                                                                      uint32_t *__restrict__ total_out,
                                                                      const uint32_t *__restrict__ skip,
                                                                      const uint32_t *__restrict__ live, uint32_t num_blocks) {
    __shared__ uint32_t smem[downsweep_lds_words(KS, 2)];
    uint32_t total;
    const uint32_t my_digit_base = block_exclusive_scan(digit_total[threadIdx.x], smem + DS_WAVE_TOT, &total);
    if (FIRST) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *total_out = total;  // V: the splats that emit pairs this frame
        // (live: [1] = number of live partitions, the list behind the block list — projection.hip LIVE_HEADER = 8)
        downsweep_partitions<KS, 2, true, 8, ATOMIC_RANK>(io, host_count, shift, part_hist, stride,
                                                    (uint32_t)(KS * SORT_BLOCK / PROJ_BLOCK), my_digit_base, smem, skip,
                                                    live != nullptr ? live + 8 + num_blocks : nullptr,
                                                    live != nullptr ? live[1] : 0u);
    } else {
        const uint32_t count = *d_count;
        if (count <= small_count)
            downsweep_partitions<KPT_SMALL, 2, false, 8, ATOMIC_RANK>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
        else
            downsweep_partitions<KS, 2, false, 8, ATOMIC_RANK>(io, count, shift, part_hist, stride, 1u, my_digit_base, smem);
    }
}

// Does the LDS unit hand out same-address returning atomics of one wave instruction in ascending lane order (and those
// of successive instructions in program order)?  One workgroup ranks 16 rounds of digit patterns both ways — the
// returning atomic and the ballot form — on all-equal, two-valued, strided, hashed and partially inactive rounds, and
// reports the number of lanes that disagree.
__global__ __launch_bounds__(SORT_BLOCK) void rank_selftest_kernel(uint32_t *__restrict__ mismatches) {
    __shared__ uint32_t cnt_a[SORT_WAVES][RADIX], cnt_b[SORT_WAVES][RADIX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < SORT_WAVES; ++w) cnt_a[w][threadIdx.x] = cnt_b[w][threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile uint32_t *my_cnt = cnt_b[wave];
    uint32_t bad = 0;
    for (int r = 0; r < 16; ++r) {
        const uint32_t h = ((uint32_t)(lane + 64 * wave) * 2654435761u + (uint32_t)r * 40503u) >> 7;
        const uint32_t d = r < 3 ? 5u : r < 5 ? (uint32_t)(lane & 1) : r < 7 ? (uint32_t)(lane % 7) : r < 9 ? (uint32_t)(lane >> 3)
                         : r < 11 ? (uint32_t)(lane & 0x30) : (h & 255u);
        const bool ok = r % 4 != 3 || (h & 0x300u) != 0u;  // every fourth round: a quarter of the lanes sit out
        uint32_t ra = 0, rb = 0;
        if (ok) ra = atomicAdd(&cnt_a[wave][d], 1u);
        unsigned long long m = __ballot(ok);
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        if (ok) {
            const uint32_t before = my_cnt[d];
            rb = before + (uint32_t)__popcll(m & lt_mask);
            if ((m >> lane) <= 1ull) my_cnt[d] = rb + 1u;
        }
        bad += (ok && ra != rb) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------------------------------------------
// The pair level in ONE pass ("wide" pass): a counting sort on the whole tile id.
//
// With 16-bit keys that are STRIPE-LOCAL tile ids (projection.hip: emit_kernel) a context whose stripe — or whole frame —
// has at most 4096 tiles needs ceil(log2(tiles)) <= 12 key bits sorted; the split form above takes two passes of <= 8 bits
// for anything above 256 tiles: six launches, two reads and two writes of every pair.  At the sizes where that matters —
// a stripe rank of an 8-GPU frame (1.2 - 1.9 M pairs), a 100 k-splat scene — every one of those launches is latency-bound
// (5 - 15 us whatever its input) and a kernel boundary is the cheapest synchronisation this chip has, so
// the lever is fewer PASSES, not fused launches: NB = 1024 or 4096 bins, three launches, one read + one write of the pairs.
//
//   wide_upsweep    per partition: NB-bin histogram in LDS -> row p of the count matrix hist[p][NB] (partition-major: a
//                   partition's row is one contiguous, coalesced 4 / 16 KiB store)
//   wide_spine      one workgroup per 64 digits, lane = digit, its 16 waves share the partitions: exclusive prefix down
//                   every column, in place; digit totals
//   wide_downsweep  per partition: digit bases (block scan of the totals) + the partition's column prefixes = running
//                   offsets in LDS; per 4096-key step the stable rank of a key among its wave's earlier keys with the same
//                   tile is ONE returning LDS atomic on per-wave counters (two 16-bit counters per word above 1024 bins:
//                   the same lane-ordered resolution the split form relies on, self-tested in this very shape), the
//                   waves' counts are turned into prefixes, and every lane scatters its own pairs — no reorder through
//                   LDS: with as many bins as a step has keys per wave a digit run is a handful of elements anyway, and
//                   the whole output of such a frame (7 - 11 MB) lives in the L2s until it is complete.
// The count matrix has at most WIDE_MAX_PARTS rows whatever the pair count: beyond WIDE_MAX_PARTS x 4096 pairs a
// partition is several 4096-key steps long and its workgroup carries the running offsets from step to step — correct for
// any count, and the host only takes this form while the previous frames' pair counts were small (api.hip).
// ---------------------------------------------------------------------------------------------------
constexpr int WIDE_KPT = 16;
constexpr uint32_t WIDE_STEP = SORT_BLOCK * WIDE_KPT;  // 4096 keys per step
struct WideGeom {
    uint32_t parts, steps;  // partition p = steps [p * steps, (p + 1) * steps) of WIDE_STEP keys
};
__device__ __host__ __forceinline__ WideGeom wide_geom(uint32_t count) {
    const uint32_t total_steps = (count + WIDE_STEP - 1u) / WIDE_STEP;
    WideGeom g;
    g.steps = (total_steps + WIDE_MAX_PARTS - 1u) / WIDE_MAX_PARTS;
    if (g.steps == 0u) g.steps = 1u;
    g.parts = (total_steps + g.steps - 1u) / g.steps;
    return g;
}

template <int NB>
__global__ __launch_bounds__(SORT_BLOCK) void wide_upsweep_kernel(const uint16_t *__restrict__ keys,
                                                                  const uint32_t *__restrict__ d_count,
                                                                  uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[NB];
    const uint32_t count = *d_count;
    const WideGeom g = wide_geom(count);
    GSPLAT_FOR_PARTITIONS(p, g.parts) {
#pragma unroll
        for (int j = 0; j < NB / SORT_BLOCK; ++j) h[j * SORT_BLOCK + threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t begin = p * g.steps * WIDE_STEP;
        const uint32_t end = min(count, begin + g.steps * WIDE_STEP);
        for (uint32_t base = begin; base < end; base += WIDE_STEP) {
            if (base + WIDE_STEP <= end) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {  // 16 keys per lane as two 16-byte loads
                    const uint4 k = reinterpret_cast<const uint4 *>(keys + base)[i * SORT_BLOCK + threadIdx.x];
                    const uint32_t w[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        atomicAdd(&h[(w[e] & 0xFFFFu) & (NB - 1)], 1u);
                        atomicAdd(&h[(w[e] >> 16) & (NB - 1)], 1u);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < WIDE_KPT; ++i) {
                    const uint32_t idx = base + i * SORT_BLOCK + threadIdx.x;
                    if (idx < end) atomicAdd(&h[keys[idx] & (NB - 1)], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t *row = hist + (size_t)p * NB;
#pragma unroll
        for (int j = 0; j < NB / SORT_BLOCK; ++j) row[j * SORT_BLOCK + threadIdx.x] = h[j * SORT_BLOCK + threadIdx.x];
        __syncthreads();
    }
}

// grid NB / 64; 1024 lanes: wave w takes the w-th sixteenth of the partitions, lane l digit 64 * blockIdx.x + l.  A lane's
// share of its column (at most WIDE_MAX_PARTS / 16 = 64 counts) stays in registers between the sum and the scan: one
// round trip to memory, then the stores (the first version re-read the column from the L2 for its second sweep).
template <int NB>
__global__ __launch_bounds__(SPINE_BLOCK) void wide_spine_kernel(uint32_t *__restrict__ hist,
                                                                 const uint32_t *__restrict__ d_count,
                                                                 uint32_t *__restrict__ digit_total) {
    constexpr int MAX_PER = (int)(WIDE_MAX_PARTS / (SPINE_BLOCK / 64));
    static_assert(MAX_PER == 64, "a lane keeps its share of a column in 64 registers");
    __shared__ uint32_t chunk_sum[SPINE_BLOCK / 64][64];
    const uint32_t parts = wide_geom(*d_count).parts;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d = blockIdx.x * 64u + (uint32_t)lane;
    const uint32_t per = (parts + 15u) / 16u;
    const uint32_t p0 = min(parts, (uint32_t)wave * per), p1 = min(parts, p0 + per);
    uint32_t v[MAX_PER];
    uint32_t sum = 0;
#pragma unroll
    for (int c = 0; c < MAX_PER; c += 16) {
        if (p0 + (uint32_t)c < p1) {  // (wave-uniform: a short column costs the loads it has)
#pragma unroll
            for (int k = c; k < c + 16; ++k) v[k] = p0 + (uint32_t)k < p1 ? hist[(size_t)(p0 + k) * NB + d] : 0u;
        } else {
#pragma unroll
            for (int k = c; k < c + 16; ++k) v[k] = 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < MAX_PER; ++k) sum += v[k];
    chunk_sum[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SPINE_BLOCK / 64; ++w) {
        const uint32_t c = chunk_sum[w][lane];
        run += w < wave ? c : 0u;
        total += c;
    }
#pragma unroll
    for (int c = 0; c < MAX_PER; c += 16) {
        if (p0 + (uint32_t)c < p1) {
#pragma unroll
            for (int k = c; k < c + 16; ++k) {
                if (p0 + (uint32_t)k < p1) hist[(size_t)(p0 + k) * NB + d] = run;
                run += v[k];
            }
        }
    }
    if (wave == 0) digit_total[d] = total;
}

// per-wave counters: u32 each up to 1024 bins, two u16 per word above (a step gives a wave at most 1024 keys)
template <int NB>
struct WideCounters {
    static constexpr bool PACKED = NB > 1024;
    static constexpr int WORDS = PACKED ? NB / 2 : NB;
    // the returning atomic: rank of this key among the wave's earlier keys of the step with the same digit
    __device__ static __forceinline__ uint32_t take(uint32_t *row, uint32_t d) {
        if constexpr (PACKED) {
            const uint32_t sh = (d & 1u) * 16u;
            return (atomicAdd(&row[d >> 1], 1u << sh) >> sh) & 0xFFFFu;
        } else {
            return atomicAdd(&row[d], 1u);
        }
    }
    __device__ static __forceinline__ uint32_t read(const uint32_t *row, uint32_t d) {
        if constexpr (PACKED) return (row[d >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu;
        else return row[d];
    }
};

template <int NB>
__global__ __launch_bounds__(SORT_BLOCK) void wide_downsweep_kernel(const uint16_t *__restrict__ key_in,
                                                                    const uint32_t *__restrict__ val_in,
                                                                    uint16_t *__restrict__ key_out,
                                                                    uint32_t *__restrict__ val_out,
                                                                    const uint32_t *__restrict__ d_count,
                                                                    const uint32_t *__restrict__ hist,
                                                                    const uint32_t *__restrict__ digit_total) {
    using C = WideCounters<NB>;
    constexpr int DPT = NB / SORT_BLOCK;  // digits per lane in the per-digit phases (a contiguous run: 4 or 16)
    __shared__ uint32_t cnt[SORT_WAVES][C::WORDS];
    __shared__ uint32_t start[NB];  // where this step's first element of a digit goes
    __shared__ uint32_t off[NB];    // ... and the next step's (running, per partition)
    __shared__ uint32_t wave_tot[8];
    const uint32_t count = *d_count;
    if (count == 0u) return;  // (round B of a frame whose round A finished every tile)
    const WideGeom g = wide_geom(count);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // global base of every digit: exclusive scan of the pass's digit totals (identical in every workgroup)
    uint32_t gbase[DPT];
    {
        uint32_t v[DPT], mine = 0;
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            v[j] = digit_total[threadIdx.x * DPT + j];
            mine += v[j];
        }
        uint32_t unused;
        uint32_t run = block_exclusive_scan(mine, wave_tot, &unused);
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            gbase[j] = run;
            run += v[j];
        }
    }
    GSPLAT_FOR_PARTITIONS(p, g.parts) {
        const uint32_t *row = hist + (size_t)p * NB;
#pragma unroll
        for (int j = 0; j < DPT; ++j) off[threadIdx.x * DPT + j] = gbase[j] + row[threadIdx.x * DPT + j];
        const uint32_t begin = p * g.steps * WIDE_STEP;
        const uint32_t end = min(count, begin + g.steps * WIDE_STEP);
        for (uint32_t base = begin; base < end; base += WIDE_STEP) {
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w)
#pragma unroll
                for (int j = 0; j < C::WORDS / SORT_BLOCK; ++j) cnt[w][j * SORT_BLOCK + threadIdx.x] = 0u;
            // wave-striped: a wave's 1024 keys are consecutive in the input, so "earlier" = (lower wave, lower r, lower lane)
            const uint32_t wbase = base + (uint32_t)wave * (WIDE_KPT * 64u) + (uint32_t)lane;
            const bool full = base + WIDE_STEP <= end;
            uint32_t key[WIDE_KPT], val[WIDE_KPT], rank[WIDE_KPT];
#pragma unroll
            for (int r = 0; r < WIDE_KPT; ++r) {
                const uint32_t idx = wbase + r * 64u;
                const bool ok = full || idx < end;
                key[r] = ok ? ((uint32_t)key_in[idx] & (NB - 1)) : 0xFFFFFFFFu;
                val[r] = ok ? val_in[idx] : 0u;
            }
            __syncthreads();  // counters zeroed (and, first step, the running offsets in place)
#pragma unroll
            for (int r = 0; r < WIDE_KPT; ++r)
                if (key[r] != 0xFFFFFFFFu) rank[r] = C::take(cnt[wave], key[r]);
            __syncthreads();
            // per digit: the waves' counts become exclusive prefixes, the running offset moves on
            if constexpr (C::PACKED) {
#pragma unroll
                for (int j = 0; j < DPT / 2; ++j) {  // one word = two digits
                    const uint32_t wd = threadIdx.x * (DPT / 2) + j;
                    uint32_t run = 0;
#pragma unroll
                    for (int w = 0; w < SORT_WAVES; ++w) {
                        const uint32_t c = cnt[w][wd];
                        cnt[w][wd] = run;  // (a step has 4096 keys: no half ever reaches 2^16, nothing carries into its neighbour)
                        run += c;
                    }
                    const uint32_t d0 = 2u * wd, d1 = d0 + 1u;
                    const uint32_t o0 = off[d0], o1 = off[d1];
                    start[d0] = o0; start[d1] = o1;
                    off[d0] = o0 + (run & 0xFFFFu); off[d1] = o1 + (run >> 16);
                }
            } else {
#pragma unroll
                for (int j = 0; j < DPT; ++j) {
                    const uint32_t d = threadIdx.x * DPT + j;
                    uint32_t run = 0;
#pragma unroll
                    for (int w = 0; w < SORT_WAVES; ++w) {
                        const uint32_t c = cnt[w][d];
                        cnt[w][d] = run;
                        run += c;
                    }
                    const uint32_t o = off[d];
                    start[d] = o;
                    off[d] = o + run;
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < WIDE_KPT; ++r) {
                if (key[r] != 0xFFFFFFFFu) {
                    const uint32_t dst = start[key[r]] + C::read(cnt[wave], key[r]) + rank[r];
                    key_out[dst] = (uint16_t)key[r];
                    val_out[dst] = val[r];
                }
            }
            __syncthreads();  // (the next step zeroes the counters; `start` is rewritten after its ranking)
        }
    }
}

// The packed counters of the wide pass rest on the same property of the LDS unit as the split form's (same-address lanes
// of one returning atomic are served in ascending lane order, a wave's instructions in program order) — with per-lane
// ADDENDS that differ (1 or 1 << 16).  Checked in that shape: 16 rounds of 12-bit digits, ranks against the ballot form.
__global__ __launch_bounds__(SORT_BLOCK) void rank_selftest_packed_kernel(uint32_t *__restrict__ mismatches) {
    using C = WideCounters<4096>;
    __shared__ uint32_t cnt_a[SORT_WAVES][C::WORDS];
    __shared__ uint16_t cnt_b[SORT_WAVES][4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < SORT_WAVES; ++w) {
        for (int j = threadIdx.x; j < C::WORDS; j += SORT_BLOCK) cnt_a[w][j] = 0u;
        for (int j = threadIdx.x; j < 4096; j += SORT_BLOCK) cnt_b[w][j] = 0;
    }
    __syncthreads();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    volatile uint16_t *my_cnt = cnt_b[wave];
    uint32_t bad = 0;
    for (int r = 0; r < 16; ++r) {
        const uint32_t hsh = ((uint32_t)(lane + 64 * wave) * 2654435761u + (uint32_t)r * 40503u) >> 7;
        // all equal; the two halves of ONE word alternating; neighbours sharing words; strided; hashed over few / all bins
        const uint32_t d = r < 2 ? 77u : r < 5 ? (uint32_t)(2 * 19 + (lane & 1)) : r < 7 ? (uint32_t)(lane % 6)
                         : r < 9 ? (uint32_t)(lane >> 2) * 2u + 1u : r < 12 ? (hsh & 15u) + 4000u : (hsh & 4095u);
        const bool ok = r % 4 != 3 || (hsh & 0x300u) != 0u;
        uint32_t ra = 0, rb = 0;
        if (ok) ra = C::take(cnt_a[wave], d);
        unsigned long long m = __ballot(ok);
        for (int b = 0; b < 12; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        if (ok) {
            const uint32_t before = my_cnt[d];
            rb = before + (uint32_t)__popcll(m & lt_mask);
            if ((m >> lane) <= 1ull) my_cnt[d] = (uint16_t)(rb + 1u);
        }
        bad += (ok && ra != rb) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
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

bool sort_rank_selftest() {  // on the current device
    uint32_t *d_bad = nullptr, h_bad = ~0u;
    if (hipMalloc(&d_bad, sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    bool ok = hipMemset(d_bad, 0, sizeof(uint32_t)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(rank_selftest_kernel, dim3(64), dim3(SORT_BLOCK), 0, nullptr, d_bad);
        hipLaunchKernelGGL(rank_selftest_packed_kernel, dim3(64), dim3(SORT_BLOCK), 0, nullptr, d_bad);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpy(&h_bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess && h_bad == 0u;
    }
    (void)hipFree(d_bad);
    if (!ok) (void)hipGetLastError();  // a failed self-test means "use the ballot form", not a sticky error for the caller
    return ok;
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

namespace {
template <int KS>
void sort_splats_with(SortBuffers &sb, const SplatKeys &keys, uint32_t n, const uint32_t *block_skip, hipStream_t s,
                      const uint32_t *live) {
    constexpr uint32_t PART = KS * SORT_BLOCK;
    const uint32_t stride = (n + PROJ_BLOCK - 1) / PROJ_BLOCK;  // row length of splat_hist: one entry per projection workgroup
    const uint32_t small = sb.small_count;
    // pass 0 (depth16 & 255): histograms by the projection kernel; compaction of the visible splats
    hipLaunchKernelGGL(spine_kernel, dim3(RADIX), dim3(SPINE_BLOCK), 0, s, sb.splat_hist,
                       static_cast<const uint32_t *>(nullptr), stride, sb.digit_base, stride, 0u, PART, block_skip);
    SortIO<2> io0{};
    io0.key_in = keys.key; io0.pay_in[0] = nullptr; io0.pay_in[1] = keys.dims;
    io0.key_out = sb.list[1].key; io0.pay_out[0] = sb.list[1].id; io0.pay_out[1] = sb.list[1].dims;
    const uint32_t parts0 = (n + PART - 1) / PART;
    const auto first_pass = sb.rank_atomic ? downsweep_splats_kernel<true, true, KS> : downsweep_splats_kernel<true, false, KS>;
    hipLaunchKernelGGL(first_pass, dim3(grid_for(parts0)), dim3(SORT_BLOCK), 0, s, io0,
                       static_cast<const uint32_t *>(nullptr), n, 0, sb.splat_hist, sb.digit_base, stride, 0u,
                       sb.v_count, block_skip, block_skip != nullptr ? live : nullptr, stride);
    // pass 1 (depth16 >> 8) over the compact list
    const uint32_t parts1 = (n + SORT_BLOCK * KPT_SMALL - 1) / (SORT_BLOCK * KPT_SMALL);
    hipLaunchKernelGGL((upsweep_kernel<KS, uint32_t>), dim3(grid_for(parts1)), dim3(SORT_BLOCK), 0, s, sb.list[1].key,
                       sb.v_count, 8, (uint32_t)(RADIX - 1), sb.splat_hist, stride, small);
    hipLaunchKernelGGL(spine_kernel, dim3(RADIX), dim3(SPINE_BLOCK), 0, s, sb.splat_hist, sb.v_count, 0u,
                       sb.digit_base, stride, small, PART, static_cast<const uint32_t *>(nullptr));
    SortIO<2> io1{};
    io1.key_in = sb.list[1].key; io1.pay_in[0] = sb.list[1].id; io1.pay_in[1] = sb.list[1].dims;
    io1.key_out = sb.list[0].key; io1.pay_out[0] = sb.list[0].id; io1.pay_out[1] = sb.list[0].dims;
    const auto second_pass = sb.rank_atomic ? downsweep_splats_kernel<false, true, KS> : downsweep_splats_kernel<false, false, KS>;
    hipLaunchKernelGGL(second_pass, dim3(grid_for(parts1)), dim3(SORT_BLOCK), 0, s, io1,
                       sb.v_count, 0u, 8, sb.splat_hist, sb.digit_base, stride, small,
                       static_cast<uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr),
                       static_cast<const uint32_t *>(nullptr), 0u);
}
}  // namespace

static int splat_partitions_pinned() {  // GSPLAT_SPLAT_PARTITIONS=small|big pins the choice: A/B and tests; same sorted list
    static const int pinned = [] {
        const char *e = getenv("GSPLAT_SPLAT_PARTITIONS");
        return e && !strcmp(e, "small") ? 1 : (e && !strcmp(e, "big") ? 2 : 0);
    }();
    return pinned;
}
static bool splat_partitions_big(uint32_t n) {
    const int pinned = splat_partitions_pinned();
    return pinned ? pinned == 2 : n >= SPLAT_BIG_N;
}
uint32_t sort_splat_part_blocks(uint32_t n) {
    return (uint32_t)((splat_partitions_big(n) ? KPT_SPLAT_BIG : KPT_SPLAT) * SORT_BLOCK / PROJ_BLOCK);
}

void launch_sort_splats(SortBuffers &sb, const SplatKeys &keys, uint32_t n, const uint32_t *block_skip, hipStream_t s,
                        KernelTimer *kt, const uint32_t *live) {
    if (n == 0) {
        (void)hipMemsetAsync(sb.v_count, 0, sizeof(uint32_t), s);
        return;
    }
    if (splat_partitions_big(n)) sort_splats_with<KPT_SPLAT_BIG>(sb, keys, n, block_skip, s, live);
    else sort_splats_with<KPT_SPLAT>(sb, keys, n, block_skip, s, live);
    if (kt) kt->mark(GSPLAT_KERNEL_SPLAT_SORT);
}

namespace {

template <typename KeyT>
int sort_pairs_typed(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int first_shift, int total_bits,
                     hipStream_t s, KernelTimer *kt) {
    // the bits [first_shift, first_shift + total_bits) in the fewest passes of at most 8 bits, spread evenly (13 tile
    // bits = 7 + 6: fewer ballots per key and longer digit runs than 8 + 5)
    const int total = total_bits > 0 ? total_bits : 0;
    const int passes = total ? sort_num_passes(total) : 0;
    const uint32_t max_parts = sort_max_partitions(capacity);
    const uint32_t grid = grid_for(max_parts);
    constexpr int KEY_BITS = 8 * (int)sizeof(KeyT);
    int cur = 0, shift = first_shift;
    for (int pass = 0; pass < passes; ++pass) {
        int bits = total / passes + (pass < total % passes ? 1 : 0);
        if (bits < 4) bits = 4;  // (key bits above the significant ones are zero: a wider digit is the same digit)
        if (shift + bits > KEY_BITS) bits = KEY_BITS - shift;
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL((upsweep_kernel<KPT, KeyT>), dim3(grid), dim3(SORT_BLOCK), 0, s,
                           reinterpret_cast<const KeyT *>(sb.keys[cur]), d_count, shift, mask, sb.part_hist, max_parts,
                           sb.small_count);
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_UPSWEEP);
        hipLaunchKernelGGL(spine_kernel, dim3(mask + 1u), dim3(SPINE_BLOCK), 0, s, sb.part_hist, d_count, 0u,
                           sb.digit_base, max_parts, sb.small_count, (uint32_t)(SORT_BLOCK * KPT),
                           static_cast<const uint32_t *>(nullptr));
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_SPINE);
        SortIO<1, KeyT> io{};
        io.key_in = reinterpret_cast<const KeyT *>(sb.keys[cur]); io.pay_in[0] = sb.values[cur];
        io.key_out = reinterpret_cast<KeyT *>(sb.keys[cur ^ 1]); io.pay_out[0] = sb.values[cur ^ 1];
#define GSPLAT_LAUNCH_D(B)                                                                                         \
    hipLaunchKernelGGL((sb.rank_atomic ? downsweep_pairs_kernel<B, true, KeyT> : downsweep_pairs_kernel<B, false, KeyT>), \
                       dim3(grid), dim3(SORT_BLOCK), 0, s, io, d_count, shift, sb.part_hist, sb.digit_base, max_parts,   \
                       sb.small_count)
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

}  // namespace

uint32_t sort_wide_bins(uint32_t tiles) {  // bins of the one-pass form for that many (stripe-local) tile ids; 0: none
    if (tiles <= 256u || tiles > 4096u) return 0u;  // (up to 256 tiles the split form is one pass as well)
    return tiles <= 1024u ? 1024u : 4096u;
}
size_t sort_wide_hist_words(uint32_t bins) { return (size_t)WIDE_MAX_PARTS * bins + bins; }  // the count matrix + digit totals

int launch_sort_pairs_wide(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, uint32_t bins, hipStream_t s,
                           KernelTimer *kt) {
    const uint64_t max_steps = (capacity + WIDE_STEP - 1) / WIDE_STEP;
    const uint32_t grid = grid_for(max_steps < WIDE_MAX_PARTS ? max_steps : WIDE_MAX_PARTS);
    uint32_t *hist = sb.wide_hist, *totals = sb.wide_hist + (size_t)WIDE_MAX_PARTS * bins;
    const uint16_t *kin = reinterpret_cast<const uint16_t *>(sb.keys[0]);
    uint16_t *kout = reinterpret_cast<uint16_t *>(sb.keys[1]);
    if (bins == 1024u) {
        hipLaunchKernelGGL(wide_upsweep_kernel<1024>, dim3(grid), dim3(SORT_BLOCK), 0, s, kin, d_count, hist);
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_UPSWEEP);
        hipLaunchKernelGGL(wide_spine_kernel<1024>, dim3(1024 / 64), dim3(SPINE_BLOCK), 0, s, hist, d_count, totals);
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_SPINE);
        hipLaunchKernelGGL(wide_downsweep_kernel<1024>, dim3(grid), dim3(SORT_BLOCK), 0, s, kin, sb.values[0], kout,
                           sb.values[1], d_count, hist, totals);
    } else {
        hipLaunchKernelGGL(wide_upsweep_kernel<4096>, dim3(grid), dim3(SORT_BLOCK), 0, s, kin, d_count, hist);
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_UPSWEEP);
        hipLaunchKernelGGL(wide_spine_kernel<4096>, dim3(4096 / 64), dim3(SPINE_BLOCK), 0, s, hist, d_count, totals);
        if (kt) kt->mark(GSPLAT_KERNEL_SORT_SPINE);
        hipLaunchKernelGGL(wide_downsweep_kernel<4096>, dim3(grid), dim3(SORT_BLOCK), 0, s, kin, sb.values[0], kout,
                           sb.values[1], d_count, hist, totals);
    }
    if (kt) kt->mark(GSPLAT_KERNEL_SORT_DOWNSWEEP);
    return 1;
}

int launch_sort_pairs(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int sig_bits, hipStream_t s,
                      KernelTimer *kt, int first_bit, bool narrow_keys) {
    // narrow_keys: keys[] hold 16-bit tile ids (bit 16 of the reference's key = bit 0 here)
    const int total = sig_bits > first_bit ? sig_bits - first_bit : 0;
    if (narrow_keys) return sort_pairs_typed<uint16_t>(sb, d_count, capacity, first_bit - 16, total, s, kt);
    return sort_pairs_typed<uint32_t>(sb, d_count, capacity, first_bit, total, s, kt);
}

}  // namespace gsplat
