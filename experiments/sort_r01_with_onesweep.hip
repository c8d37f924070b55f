// Stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 (wave64) — replaces
// resources/shaders/compute/radix_sort_{upsweep,spine,downsweep}.glsl.
//
// The reference's sort only works with 32-wide subgroups (radix_sort_spine.glsl:33,56-59,
// radix_sort_downsweep.glsl:144-156); what is kept is its CONTRACT: after the passes the pairs are
// ascending by key and equal keys keep their input order (radix_sort_downsweep.glsl:178-213:
// dst = global[digit] + partition[digit] + local rank).  Mechanics are native wave64:
//   upsweep   : per 4096-key partition, 256-bin digit histogram in LDS (uint4 key loads)
//   spine     : one workgroup per digit, exclusive scan over partitions (+ digit totals)
//   downsweep : wave-striped key loads, match-any ranking with 8 x 64-bit ballots per key, per-wave
//               digit counters in LDS, workgroup scan, reorder through LDS, coalesced scatter of keys
//               and values in digit runs.
// The pair count lives in device memory (written by the projection pass); grids are fixed and
// partitions are grid-strided, so there is no host read-back and no indirect dispatch
// (gaussian_splatting_rasterizer.gd:146-148 used dispatch_indirect for the same reason).
// Only ceil(sig_bits/8) passes run: keys are (tile << 16 | depth16) and tile < 2^ceil(log2 T).
#include "gsplat_internal.h"

namespace gsplat {

namespace {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr int SORT_BLOCK = 256;                 // 4 wave64
constexpr int SORT_WAVES = SORT_BLOCK / 64;
#ifndef GSPLAT_SORT_KPT
#define GSPLAT_SORT_KPT 16
#endif
constexpr int KPT = GSPLAT_SORT_KPT;            // keys per lane
constexpr int PART = SORT_BLOCK * KPT;          // 4096 keys per partition
constexpr int WAVE_KEYS = PART / SORT_WAVES;    // 1024 keys per wave
constexpr uint32_t PAD_KEY = 0xFFFFFFFFu;       // radix_sort_upsweep.glsl:53
constexpr int SORT_GRID = 2048;                 // 256 CUs x 8 workgroups
// Small inputs (a stripe of an 8-GPU shard, a 100 k-splat scene) are latency-bound: with 4096-key partitions a
// 0.5 M-pair pass is 128 workgroups that each walk 16 ranking rounds.  Up to SMALL_COUNT the same kernels cut the input
// into 1024-key partitions (4 keys per lane): four times the workgroups, each a quarter as long.  The choice is made
// on the device from the pair count, identically in the three kernels of a pass; the result is the same either way.
constexpr int KPT_SMALL = 4;
constexpr int PART_SMALL = SORT_BLOCK * KPT_SMALL;
constexpr uint32_t SMALL_COUNT = 5u << 18;  // 1.3 M pairs: measured crossover (tools/sort_small_sweep.py: 0.6 M pairs -16 %, 1.2 M -5 %, 1.8 M +12 %)
__device__ __host__ __forceinline__ uint32_t partitions_of(uint32_t count, uint32_t small_count) {
    return count <= small_count ? (count + PART_SMALL - 1) / PART_SMALL : (count + PART - 1) / PART;
}

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift) { return (key >> shift) & (RADIX - 1); }

// part_hist is digit-major, part_hist[digit * max_parts + partition]: the spine scans contiguous rows.
constexpr int UPSWEEP_COPIES = 2;  // sub-histograms: spread the same-address LDS atomics of hot digits
template <int K>
__device__ __forceinline__ void upsweep_partitions(const uint32_t *__restrict__ keys, uint32_t count, int shift,
                                                   uint32_t *__restrict__ part_hist, uint32_t max_parts,
                                                   uint32_t (*hist)[RADIX]) {
    constexpr uint32_t P = SORT_BLOCK * K;
    const uint32_t num_parts = (count + P - 1) / P;
    uint32_t *my = hist[threadIdx.x & (UPSWEEP_COPIES - 1)];
    for (uint32_t p = blockIdx.x; p < num_parts; p += gridDim.x) {
#pragma unroll
        for (int c = 0; c < UPSWEEP_COPIES; ++c) hist[c][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t start = p * P;
        if (start + P <= count) {
            const uint4 *src = reinterpret_cast<const uint4 *>(keys + start);
#pragma unroll
            for (int i = 0; i < K / 4; ++i) {
                const uint4 k = src[i * SORT_BLOCK + threadIdx.x];
                atomicAdd(&my[digit_of(k.x, shift)], 1u);
                atomicAdd(&my[digit_of(k.y, shift)], 1u);
                atomicAdd(&my[digit_of(k.z, shift)], 1u);
                atomicAdd(&my[digit_of(k.w, shift)], 1u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const uint32_t idx = start + i * SORT_BLOCK + threadIdx.x;
                const uint32_t k = idx < count ? keys[idx] : PAD_KEY;
                atomicAdd(&my[digit_of(k, shift)], 1u);
            }
        }
        __syncthreads();
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < UPSWEEP_COPIES; ++c) v += hist[c][threadIdx.x];
        part_hist[(size_t)threadIdx.x * max_parts + p] = v;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_BLOCK) void upsweep_kernel(const uint32_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ d_count, int shift,
                                                             uint32_t *__restrict__ part_hist, uint32_t max_parts,
                                                             uint32_t small_count) {
    __shared__ uint32_t hist[UPSWEEP_COPIES][RADIX];
    const uint32_t count = *d_count;
    if (count <= small_count) upsweep_partitions<KPT_SMALL>(keys, count, shift, part_hist, max_parts, hist);
    else upsweep_partitions<KPT>(keys, count, shift, part_hist, max_parts, hist);
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

// One 1024-lane workgroup per digit: in-place exclusive scan of part_hist[.][digit] over partitions and
// digit_total[digit].  Each lane takes SPINE_ITEMS consecutive partitions per trip (4096 partitions = 16.7 M pairs per
// trip), so realistic frames need a single trip instead of ten.
constexpr int SPINE_BLOCK = 1024;
constexpr int SPINE_ITEMS = 4;
__global__ __launch_bounds__(SPINE_BLOCK) void spine_kernel(uint32_t *__restrict__ part_hist_all,
                                                            const uint32_t *__restrict__ d_count,
                                                            uint32_t *__restrict__ digit_total, uint32_t max_parts,
                                                            uint32_t small_count) {
    __shared__ uint32_t wave_tot[SPINE_BLOCK / 64];
    const uint32_t count = *d_count;
    const uint32_t num_parts = partitions_of(count, small_count);
    const uint32_t digit = blockIdx.x;
    uint32_t *part_hist = part_hist_all + (size_t)digit * max_parts;  // this digit's row
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

struct DownsweepShared {
    uint32_t wave_cnt[SORT_WAVES][RADIX];  // per-wave digit counters -> exclusive wave prefixes
    uint32_t local_start[RADIX];           // exclusive scan of the partition's digit counts
    uint32_t dst_base[RADIX];              // global base of each digit run minus local_start
    uint32_t wave_tot[SORT_WAVES];
    uint32_t lkeys[PART];
    uint32_t lvals[PART];
};

template <int K>
__device__ __forceinline__ void downsweep_partitions(const uint32_t *__restrict__ keys_in,
                                                     const uint32_t *__restrict__ vals_in,
                                                     uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
                                                     uint32_t count, int shift, const uint32_t *__restrict__ part_hist,
                                                     uint32_t my_digit_base, uint32_t max_parts, DownsweepShared &sh) {
    constexpr uint32_t P = SORT_BLOCK * K;
    constexpr uint32_t WK = K * 64;  // keys per wave
    const uint32_t num_parts = (count + P - 1) / P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (uint32_t p = blockIdx.x; p < num_parts; p += gridDim.x) {
        const uint32_t start = p * P;
        const uint32_t valid = min(P, count - start);
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) sh.wave_cnt[w][threadIdx.x] = 0;

        uint32_t key[K], rank[K];
        const uint32_t wbase = start + wave * WK + lane;
        const bool full = start + P <= count;
        if (full) {
#pragma unroll
            for (int r = 0; r < K; ++r) key[r] = keys_in[wbase + r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < K; ++r) {
                const uint32_t idx = wbase + r * 64;
                key[r] = idx < count ? keys_in[idx] : PAD_KEY;
            }
        }
        __syncthreads();  // counters zeroed

        // rank each key among this wave's earlier keys with the same digit (stable).  The counters are
        // re-read every round through a volatile pointer: other lanes of the wave update them.
        volatile uint32_t *my_cnt = sh.wave_cnt[wave];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const uint32_t d = digit_of(key[r], shift);
            unsigned long long m = ~0ull;
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t before = my_cnt[d];
            const uint32_t in_group = (uint32_t)__popcll(m & lt_mask);
            const bool last = (m >> lane) <= 1ull;  // highest lane of the group
            rank[r] = before + in_group;
            if (last) my_cnt[d] = before + in_group + 1u;
        }
        __syncthreads();

        // digit = threadIdx.x: wave-exclusive prefixes, partition digit count, scan over digits
        {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) {
                const uint32_t c = sh.wave_cnt[w][threadIdx.x];
                sh.wave_cnt[w][threadIdx.x] = run;
                run += c;
            }
            uint32_t tot;
            const uint32_t ls = block_exclusive_scan(run, sh.wave_tot, &tot);
            sh.local_start[threadIdx.x] = ls;
            sh.dst_base[threadIdx.x] = my_digit_base + part_hist[(size_t)threadIdx.x * max_parts + p] - ls;
        }
        __syncthreads();

        // reorder through LDS so that each digit run leaves as contiguous, coalesced stores.  The values are loaded
        // only now (not before the ranking): fewer live registers through the ballot loops.
        uint32_t val[K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const uint32_t idx = wbase + r * 64;
            val[r] = (full || idx < count) ? vals_in[idx] : 0u;
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const uint32_t d = digit_of(key[r], shift);
            const uint32_t pos = sh.local_start[d] + sh.wave_cnt[wave][d] + rank[r];
            sh.lkeys[pos] = key[r];
            sh.lvals[pos] = val[r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const uint32_t li = i * SORT_BLOCK + threadIdx.x;
            if (li < valid) {  // padding keys sort to the tail of the partition and are dropped
                const uint32_t k = sh.lkeys[li];
                const uint32_t dst = sh.dst_base[digit_of(k, shift)] + li;
                keys_out[dst] = k;
                vals_out[dst] = sh.lvals[li];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_BLOCK) void downsweep_kernel(const uint32_t *__restrict__ keys_in,
                                                               const uint32_t *__restrict__ vals_in,
                                                               uint32_t *__restrict__ keys_out,
                                                               uint32_t *__restrict__ vals_out,
                                                               const uint32_t *__restrict__ d_count, int shift,
                                                               const uint32_t *__restrict__ part_hist,
                                                               const uint32_t *__restrict__ digit_total,
                                                               uint32_t max_parts, uint32_t small_count) {
    __shared__ DownsweepShared sh;
    const uint32_t count = *d_count;
    // exclusive scan of the pass's global digit histogram (identical in every workgroup)
    uint32_t unused;
    const uint32_t my_digit_base = block_exclusive_scan(digit_total[threadIdx.x], sh.wave_tot, &unused);
    if (count <= small_count)
        downsweep_partitions<KPT_SMALL>(keys_in, vals_in, keys_out, vals_out, count, shift, part_hist, my_digit_base,
                                        max_parts, sh);
    else
        downsweep_partitions<KPT>(keys_in, vals_in, keys_out, vals_out, count, shift, part_hist, my_digit_base,
                                  max_parts, sh);
}


// ---------------------------------------------------------------------------------------------------
// Onesweep variant (default): one histogram kernel per frame + ONE kernel per pass.
//   histogram_kernel : reads the keys once, builds the 256-bin histogram of every pass's digit (LDS atomics,
//                      one global atomicAdd per bin and workgroup) and clears the look-back state of the
//                      partitions this frame will use.
//   onesweep_kernel  : a workgroup takes the next partition from a ticket counter (so every predecessor is
//                      already running: forward progress without co-residency assumptions), ranks its 4096 keys
//                      exactly like downsweep_kernel, publishes its per-digit counts, and obtains the exclusive
//                      prefix over earlier partitions by decoupled look-back: lane d polls status[q][d] of the
//                      preceding partitions until it meets an inclusive prefix.
// Inter-workgroup protocol (cdna_hip_programming.md G16, form R2): every status word is ONE 32-bit granule
// {2 flag bits | 30-bit count} written with a relaxed agent-scope atomic store and polled with relaxed agent-scope
// atomic loads (sc1: L1 bypassed) — the datum is the flag, so no fence and no ordering between words is needed.
// Keys move HBM -> LDS -> HBM once per pass (16 B/pair) instead of 20 B/pair + spine.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t ST_FLAG_AGG = 1u << 30;
constexpr uint32_t ST_FLAG_INC = 2u << 30;
constexpr uint32_t ST_VALUE_MASK = (1u << 30) - 1u;
constexpr uint32_t SPIN_LIMIT = 1u << 22;  // bounded spin: a protocol bug must not hang the GPU

constexpr int HIST_COPIES = 8;  // sub-histograms per workgroup: spreads the same-address LDS atomics of hot digits
__global__ __launch_bounds__(SORT_BLOCK) void histogram_kernel(const uint32_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ d_count, int passes,
                                                               uint32_t *__restrict__ global_hist /*[4][256]*/,
                                                               uint32_t *__restrict__ status, uint32_t max_parts,
                                                               uint32_t *__restrict__ tickets) {
    __shared__ uint32_t hist[HIST_COPIES][4][RADIX];
    const uint32_t count = *d_count;
    const uint32_t num_parts = (count + PART - 1) / PART;
    uint32_t *hflat = &hist[0][0][0];
    for (int i = threadIdx.x; i < HIST_COPIES * 4 * RADIX; i += SORT_BLOCK) hflat[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 4) tickets[threadIdx.x] = 0;
    // clear the look-back words of the partitions in use, for every pass
    for (int q = 0; q < passes; ++q) {
        uint4 *st = reinterpret_cast<uint4 *>(status + (size_t)q * max_parts * RADIX);
        const uint32_t n4 = num_parts * (RADIX / 4);
        for (uint32_t i = blockIdx.x * SORT_BLOCK + threadIdx.x; i < n4; i += gridDim.x * SORT_BLOCK)
            st[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    uint32_t(*my)[RADIX] = hist[threadIdx.x & (HIST_COPIES - 1)];
    const uint32_t n4 = count / 4;
    const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
    for (uint32_t i = blockIdx.x * SORT_BLOCK + threadIdx.x; i < n4; i += gridDim.x * SORT_BLOCK) {
        const uint4 k = k4[i];
        const uint32_t kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            for (int q = 0; q < passes; ++q) atomicAdd(&my[q][(kk[e] >> (8 * q)) & 255u], 1u);
    }
    if (blockIdx.x == 0) {
        const uint32_t i = n4 * 4 + threadIdx.x;  // tail (< 4 keys)
        if (i < count) {
            const uint32_t k = keys[i];
            for (int q = 0; q < passes; ++q) atomicAdd(&my[q][(k >> (8 * q)) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int q = 0; q < passes; ++q) {
        uint32_t v = 0;
#pragma unroll
        for (int c = 0; c < HIST_COPIES; ++c) v += hist[c][q][threadIdx.x];
        if (v) atomicAdd(&global_hist[q * RADIX + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(SORT_BLOCK) void onesweep_kernel(const uint32_t *__restrict__ keys_in,
                                                              const uint32_t *__restrict__ vals_in,
                                                              uint32_t *__restrict__ keys_out,
                                                              uint32_t *__restrict__ vals_out,
                                                              const uint32_t *__restrict__ d_count, int shift,
                                                              const uint32_t *__restrict__ digit_total,
                                                              uint32_t *status, uint32_t *ticket,
                                                              uint32_t *__restrict__ error_flag) {
    __shared__ uint32_t wave_cnt[SORT_WAVES][RADIX];
    __shared__ uint32_t local_start[RADIX];
    __shared__ uint32_t dst_base[RADIX];
    __shared__ uint32_t wave_tot[SORT_WAVES];
    __shared__ uint32_t lkeys[PART];
    __shared__ uint32_t lvals[PART];
    __shared__ uint32_t s_part;

    const uint32_t count = *d_count;
    const uint32_t num_parts = (count + PART - 1) / PART;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    uint32_t unused;
    const uint32_t my_digit_base = block_exclusive_scan(digit_total[threadIdx.x], wave_tot, &unused);

    for (;;) {
        if (threadIdx.x == 0) s_part = atomicAdd(ticket, 1u);
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t p = s_part;
        if (p >= num_parts) break;
        const uint32_t start = p * PART;
        const uint32_t valid = min((uint32_t)PART, count - start);

        uint32_t key[KPT], val[KPT], rank[KPT];
        const uint32_t wbase = start + wave * WAVE_KEYS + lane;
        if (start + PART <= count) {
#pragma unroll
            for (int r = 0; r < KPT; ++r) key[r] = keys_in[wbase + r * 64];
#pragma unroll
            for (int r = 0; r < KPT; ++r) val[r] = vals_in[wbase + r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < KPT; ++r) {
                const uint32_t idx = wbase + r * 64;
                key[r] = idx < count ? keys_in[idx] : PAD_KEY;
                val[r] = idx < count ? vals_in[idx] : 0u;
            }
        }

        volatile uint32_t *my_cnt = wave_cnt[wave];
#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            const uint32_t d = digit_of(key[r], shift);
            unsigned long long m = ~0ull;
#pragma unroll
            for (int b = 0; b < RADIX_BITS; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t before = my_cnt[d];
            const uint32_t in_group = (uint32_t)__popcll(m & lt_mask);
            const bool last = (m >> lane) <= 1ull;
            rank[r] = before + in_group;
            if (last) my_cnt[d] = before + in_group + 1u;
        }
        __syncthreads();

        // digit = threadIdx.x: partition count -> publish -> look back
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const uint32_t c = wave_cnt[w][threadIdx.x];
            wave_cnt[w][threadIdx.x] = run;
            run += c;
        }
        uint32_t *my_status = status + (size_t)p * RADIX + threadIdx.x;
        __hip_atomic_store(my_status, run | (p == 0 ? ST_FLAG_INC : ST_FLAG_AGG), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (p > 0) {
            // Walk back over the preceding partitions LB words at a time: the LB loads are independent (latency of
            // one L2 round trip per batch, not per word); words are consumed nearest-first and the walk stops at the
            // first inclusive prefix.  A word that is not published yet makes the lane re-poll from that word.
            constexpr int LB = 8;
            int64_t q = (int64_t)p - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                uint32_t sv[LB];
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    const int64_t qq = q - k;
                    sv[k] = qq >= 0 ? __hip_atomic_load(status + (size_t)qq * RADIX + threadIdx.x, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT)
                                    : ST_FLAG_INC;  // below partition 0: an empty inclusive prefix
                }
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    if (done) break;
                    if (sv[k] & (ST_FLAG_AGG | ST_FLAG_INC)) {
                        excl += sv[k] & ST_VALUE_MASK;
                        --q;
                        spins = 0;
                        if (sv[k] & ST_FLAG_INC) done = true;
                    } else {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT) {
                            *error_flag = 1u;
                            done = true;
                        }
                        break;  // re-poll starting at this word
                    }
                }
            }
            __hip_atomic_store(my_status, (excl + run) | ST_FLAG_INC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t tot;
        const uint32_t ls = block_exclusive_scan(run, wave_tot, &tot);
        local_start[threadIdx.x] = ls;
        dst_base[threadIdx.x] = my_digit_base + excl - ls;
        __syncthreads();

#pragma unroll
        for (int r = 0; r < KPT; ++r) {
            const uint32_t d = digit_of(key[r], shift);
            const uint32_t pos = local_start[d] + wave_cnt[wave][d] + rank[r];
            lkeys[pos] = key[r];
            lvals[pos] = val[r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t li = i * SORT_BLOCK + threadIdx.x;
            if (li < valid) {
                const uint32_t k = lkeys[li];
                const uint32_t dst = dst_base[digit_of(k, shift)] + li;
                keys_out[dst] = k;
                vals_out[dst] = lvals[li];
            }
        }
        __syncthreads();
    }
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
    const uint64_t big = (capacity + PART - 1) / PART;
    const uint64_t small_pairs = capacity < SMALL_COUNT ? capacity : SMALL_COUNT;
    const uint64_t small = (small_pairs + PART_SMALL - 1) / PART_SMALL;
    return (uint32_t)(big > small ? big : small);
}

int launch_sort_pairs(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int sig_bits, hipStream_t s,
                      KernelTimer *kt, int first_bit) {
    // first_bit > 0 (tile-major sort): only the bits [first_bit, sig_bits) are sorted, in 8-bit passes from first_bit
    if (sb.onesweep) first_bit = 0;
    const int passes = sig_bits > first_bit ? sort_num_passes(sig_bits - first_bit) : 0;
    const uint32_t max_parts = sort_max_partitions(capacity);
    const uint32_t grid = max_parts < (uint32_t)SORT_GRID ? (max_parts ? max_parts : 1u) : (uint32_t)SORT_GRID;
    int cur = 0;
    if (sb.onesweep) {
        (void)hipMemsetAsync(sb.global_hist, 0, 4 * RADIX * sizeof(uint32_t), s);
        // few workgroups: each ends with 256 x passes global atomics on the same 1 KiB, which serialise per address
        const uint32_t hgrid = grid < 512u ? grid : 512u;
        hipLaunchKernelGGL(histogram_kernel, dim3(hgrid), dim3(SORT_BLOCK), 0, s, sb.keys[0], d_count, passes,
                           sb.global_hist, sb.status, max_parts, sb.tickets);
        if (kt) kt->mark(3);
        for (int pass = 0; pass < passes; ++pass) {
            hipLaunchKernelGGL(onesweep_kernel, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.keys[cur], sb.values[cur],
                               sb.keys[cur ^ 1], sb.values[cur ^ 1], d_count, pass * RADIX_BITS,
                               sb.global_hist + pass * RADIX, sb.status + (size_t)pass * max_parts * RADIX,
                               sb.tickets + pass, sb.error_flag);
            if (kt) kt->mark(5);
            cur ^= 1;
        }
        return cur;
    }
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = first_bit + pass * RADIX_BITS;
        hipLaunchKernelGGL(upsweep_kernel, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.keys[cur], d_count, shift,
                           sb.part_hist, max_parts, sb.small_count);
        if (kt) kt->mark(3);
        hipLaunchKernelGGL(spine_kernel, dim3(RADIX), dim3(SPINE_BLOCK), 0, s, sb.part_hist, d_count, sb.digit_base,
                           max_parts, sb.small_count);
        if (kt) kt->mark(4);
        hipLaunchKernelGGL(downsweep_kernel, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.keys[cur], sb.values[cur],
                           sb.keys[cur ^ 1], sb.values[cur ^ 1], d_count, shift, sb.part_hist, sb.digit_base, max_parts,
                           sb.small_count);
        if (kt) kt->mark(5);
        cur ^= 1;
    }
    return cur;
}

}  // namespace gsplat
