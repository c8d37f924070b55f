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
constexpr int KPT = 16;                         // keys per lane
constexpr int PART = SORT_BLOCK * KPT;          // 4096 keys per partition
constexpr int WAVE_KEYS = PART / SORT_WAVES;    // 1024 keys per wave
constexpr uint32_t PAD_KEY = 0xFFFFFFFFu;       // radix_sort_upsweep.glsl:53
constexpr int SORT_GRID = 2048;                 // 256 CUs x 8 workgroups

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift) { return (key >> shift) & (RADIX - 1); }

__global__ __launch_bounds__(SORT_BLOCK) void upsweep_kernel(const uint32_t *__restrict__ keys,
                                                             const uint32_t *__restrict__ d_count, int shift,
                                                             uint32_t *__restrict__ part_hist) {
    __shared__ uint32_t hist[RADIX];
    const uint32_t count = *d_count;
    const uint32_t num_parts = (count + PART - 1) / PART;
    for (uint32_t p = blockIdx.x; p < num_parts; p += gridDim.x) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t start = p * PART;
        if (start + PART <= count) {
            const uint4 *src = reinterpret_cast<const uint4 *>(keys + start);
#pragma unroll
            for (int i = 0; i < KPT / 4; ++i) {
                const uint4 k = src[i * SORT_BLOCK + threadIdx.x];
                atomicAdd(&hist[digit_of(k.x, shift)], 1u);
                atomicAdd(&hist[digit_of(k.y, shift)], 1u);
                atomicAdd(&hist[digit_of(k.z, shift)], 1u);
                atomicAdd(&hist[digit_of(k.w, shift)], 1u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = start + i * SORT_BLOCK + threadIdx.x;
                const uint32_t k = idx < count ? keys[idx] : PAD_KEY;
                atomicAdd(&hist[digit_of(k, shift)], 1u);
            }
        }
        __syncthreads();
        part_hist[(size_t)p * RADIX + threadIdx.x] = hist[threadIdx.x];
        __syncthreads();
    }
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

// One workgroup per digit: in-place exclusive scan of part_hist[.][digit] over partitions; digit_total[digit].
__global__ __launch_bounds__(SORT_BLOCK) void spine_kernel(uint32_t *__restrict__ part_hist,
                                                           const uint32_t *__restrict__ d_count,
                                                           uint32_t *__restrict__ digit_total) {
    __shared__ uint32_t wave_tot[SORT_WAVES];
    const uint32_t count = *d_count;
    const uint32_t num_parts = (count + PART - 1) / PART;
    const uint32_t digit = blockIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < num_parts; base += SORT_BLOCK) {
        const uint32_t p = base + threadIdx.x;
        const uint32_t v = p < num_parts ? part_hist[(size_t)p * RADIX + digit] : 0u;
        uint32_t tot;
        const uint32_t excl = block_exclusive_scan(v, wave_tot, &tot);
        if (p < num_parts) part_hist[(size_t)p * RADIX + digit] = carry + excl;
        carry += tot;
    }
    if (threadIdx.x == 0) digit_total[digit] = carry;
}

__global__ __launch_bounds__(SORT_BLOCK) void downsweep_kernel(const uint32_t *__restrict__ keys_in,
                                                               const uint32_t *__restrict__ vals_in,
                                                               uint32_t *__restrict__ keys_out,
                                                               uint32_t *__restrict__ vals_out,
                                                               const uint32_t *__restrict__ d_count, int shift,
                                                               const uint32_t *__restrict__ part_hist,
                                                               const uint32_t *__restrict__ digit_total) {
    __shared__ uint32_t wave_cnt[SORT_WAVES][RADIX];  // per-wave digit counters -> exclusive wave prefixes
    __shared__ uint32_t local_start[RADIX];           // exclusive scan of the partition's digit counts
    __shared__ uint32_t dst_base[RADIX];              // global base of each digit run minus local_start
    __shared__ uint32_t wave_tot[SORT_WAVES];
    __shared__ uint32_t lkeys[PART];
    __shared__ uint32_t lvals[PART];

    const uint32_t count = *d_count;
    const uint32_t num_parts = (count + PART - 1) / PART;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // exclusive scan of the pass's global digit histogram (identical in every workgroup)
    uint32_t unused;
    const uint32_t my_digit_base = block_exclusive_scan(digit_total[threadIdx.x], wave_tot, &unused);

    for (uint32_t p = blockIdx.x; p < num_parts; p += gridDim.x) {
        const uint32_t start = p * PART;
        const uint32_t valid = min((uint32_t)PART, count - start);
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;

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
        __syncthreads();  // counters zeroed

        // rank each key among this wave's earlier keys with the same digit (stable).  The counters are
        // re-read every round through a volatile pointer: other lanes of the wave update them.
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
                const uint32_t c = wave_cnt[w][threadIdx.x];
                wave_cnt[w][threadIdx.x] = run;
                run += c;
            }
            uint32_t tot;
            const uint32_t ls = block_exclusive_scan(run, wave_tot, &tot);
            local_start[threadIdx.x] = ls;
            dst_base[threadIdx.x] = my_digit_base + part_hist[(size_t)p * RADIX + threadIdx.x] - ls;
        }
        __syncthreads();

        // reorder through LDS so that each digit run leaves as contiguous, coalesced stores
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
            if (li < valid) {  // padding keys sort to the tail of the partition and are dropped
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

uint32_t sort_max_partitions(uint64_t capacity) { return (uint32_t)((capacity + PART - 1) / PART); }

int launch_sort_pairs(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int sig_bits, hipStream_t s,
                      KernelTimer *kt) {
    const int passes = sort_num_passes(sig_bits);
    const uint32_t max_parts = sort_max_partitions(capacity);
    const uint32_t grid = max_parts < (uint32_t)SORT_GRID ? (max_parts ? max_parts : 1u) : (uint32_t)SORT_GRID;
    int cur = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = pass * RADIX_BITS;
        hipLaunchKernelGGL(upsweep_kernel, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.keys[cur], d_count, shift,
                           sb.part_hist);
        if (kt) kt->mark(3);
        hipLaunchKernelGGL(spine_kernel, dim3(RADIX), dim3(SORT_BLOCK), 0, s, sb.part_hist, d_count, sb.digit_base);
        if (kt) kt->mark(4);
        hipLaunchKernelGGL(downsweep_kernel, dim3(grid), dim3(SORT_BLOCK), 0, s, sb.keys[cur], sb.values[cur],
                           sb.keys[cur ^ 1], sb.values[cur ^ 1], d_count, shift, sb.part_hist, sb.digit_base);
        if (kt) kt->mark(5);
        cur ^= 1;
    }
    return cur;
}

}  // namespace gsplat
