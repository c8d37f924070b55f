// Per-tile depth sort for gfx950 — the second half of the tile-major pair sort ("tile-bucket +
// per-tile LDS depth sort as an alternative to the global 32-bit sort (same stable result)").
//
// The reference sorts the (tile << 16 | depth16, splat) pairs with four global 8-bit LSD passes
// (radix_sort_{upsweep,spine,downsweep}.glsl, gaussian_splatting_rasterizer.gd:144-153).  The result only has to be
// ascending by key with equal keys in emission order, and the key's two halves are independent: sorting by the tile
// bits first (two global passes of sort.hip, starting at bit 16) leaves every tile's pairs contiguous and still in
// emission order; what remains is a stable sort of each tile's segment on the low 16 bits.  That is done here by one
// workgroup per tile: a segment of up to 4096 pairs (tiles average ~1200 at 6 M splats, 1080p) is read once, goes
// through two 8-bit ranking passes — registers -> LDS -> registers -> final position — and is written once, instead
// of twice through HBM with per-partition histograms, a spine and a scatter each time.  Longer segments are streamed
// in 4096-pair chunks with running digit offsets (two passes through the other half of the ping-pong buffers).
// The ranking is sort.hip's: wave64 match-any (8 ballots per key), per-wave digit counters, workgroup scan.
#include "gsplat_internal.h"

namespace gsplat {

namespace {

constexpr int TS_RADIX = 256;
constexpr int TS_KMAX = 16;                // pairs per lane
constexpr uint32_t TS_PAD = 0xFFFFFFFFu;   // sorts behind every real pair in both passes, never written back
constexpr int TS_SMALL_WAVES = 4;          // 256 lanes: segments up to 4096 pairs (most tiles)
constexpr int TS_BIG_WAVES = 16;           // 1024 lanes, one workgroup per CU: segments up to 16384 pairs in 148 KiB of LDS
constexpr int TS_BIG_GRID = 512;

// LDS of one workgroup of W waves (the big variant needs dynamic LDS: 148 KiB)
template <int W>
struct TileSortShared {
    uint32_t wave_cnt[W][TS_RADIX];   // per-wave digit counters -> exclusive wave prefixes
    uint32_t local_start[TS_RADIX];   // exclusive scan of the chunk's digit counts
    uint32_t digit_base[TS_RADIX];    // streamed path: digit offsets inside the segment (running)
    uint32_t wave_tot[4];             // scan of the 256 digits uses the first 4 waves
    uint32_t lkeys[W * 64 * TS_KMAX];
    uint32_t lvals[W * 64 * TS_KMAX];
};

// exclusive scan over the values of lanes 0..255 (the digits); other lanes pass 0 and ignore the result
__device__ __forceinline__ uint32_t ts_digit_scan(uint32_t v, uint32_t *wave_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63 && wave < 4) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) base += wave_tot[w];
    __syncthreads();
    return base + incl - v;
}

// Stable position of every key of the chunk when ordered by digit (key >> shift) & 255.  Keys are held in
// (wave, round, lane) order = sequence order.  Leaves sh.local_start[d] = first position of digit d.
template <int W, int K>
__device__ __forceinline__ void digit_positions(const uint32_t (&key)[K], int shift, TileSortShared<W> &sh,
                                                uint32_t (&pos)[K]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int i = threadIdx.x; i < W * TS_RADIX; i += W * 64) (&sh.wave_cnt[0][0])[i] = 0;
    __syncthreads();
    volatile uint32_t *my_cnt = sh.wave_cnt[wave];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long m = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t before = my_cnt[d];
        const uint32_t in_group = (uint32_t)__popcll(m & lt_mask);
        const bool last = (m >> lane) <= 1ull;  // highest lane of the group
        pos[r] = before + in_group;
        if (last) my_cnt[d] = before + in_group + 1u;
    }
    __syncthreads();
    {   // digit = threadIdx.x < 256: wave-exclusive prefixes, digit count, scan over digits
        uint32_t run = 0;
        if (threadIdx.x < TS_RADIX) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint32_t c = sh.wave_cnt[w][threadIdx.x];
                sh.wave_cnt[w][threadIdx.x] = run;
                run += c;
            }
        }
        const uint32_t ex = ts_digit_scan(run, sh.wave_tot);
        if (threadIdx.x < TS_RADIX) sh.local_start[threadIdx.x] = ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const uint32_t d = (key[r] >> shift) & 255u;
        pos[r] += sh.local_start[d] + sh.wave_cnt[wave][d];
    }
}

// a segment of n <= 64*W*K pairs: read once, two ranking passes, written once (in place)
template <int W, int K>
__device__ __forceinline__ void sort_segment_lds(uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t s,
                                                 uint32_t n, TileSortShared<W> &sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = (uint32_t)wave * (K * 64u) + (uint32_t)lane;
    uint32_t key[K], val[K], pos[K];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const uint32_t i = first + r * 64u;
        key[r] = i < n ? keys[s + i] : TS_PAD;
        val[r] = i < n ? vals[s + i] : 0u;
    }
    digit_positions<W, K>(key, 0, sh, pos);  // depth16 low byte
#pragma unroll
    for (int r = 0; r < K; ++r) {
        sh.lkeys[pos[r]] = key[r];
        sh.lvals[pos[r]] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const uint32_t i = first + r * 64u;
        key[r] = sh.lkeys[i];
        val[r] = sh.lvals[i];
    }
    digit_positions<W, K>(key, 8, sh, pos);  // depth16 high byte (the padding keys stay behind the n real ones)
#pragma unroll
    for (int r = 0; r < K; ++r) {
        if (pos[r] < n) {
            keys[s + pos[r]] = key[r];
            vals[s + pos[r]] = val[r];
        }
    }
    __syncthreads();
}

// a longer segment: two streamed LSD passes, a -> b -> a, 64*W*16 pairs at a time in sequence order
template <int W>
__device__ __forceinline__ void sort_segment_streamed(uint32_t *__restrict__ keys_a, uint32_t *__restrict__ vals_a,
                                                      uint32_t *__restrict__ keys_b, uint32_t *__restrict__ vals_b,
                                                      uint32_t s, uint32_t n, TileSortShared<W> &sh) {
    constexpr uint32_t CHUNK = W * 64u * TS_KMAX;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = (uint32_t)wave * (TS_KMAX * 64u) + (uint32_t)lane;
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = 8 * pass;
        const uint32_t *ksrc = pass == 0 ? keys_a : keys_b, *vsrc = pass == 0 ? vals_a : vals_b;
        uint32_t *kdst = pass == 0 ? keys_b : keys_a, *vdst = pass == 0 ? vals_b : vals_a;
        // digit histogram of the whole segment -> exclusive offsets
        if (threadIdx.x < TS_RADIX) sh.digit_base[threadIdx.x] = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += W * 64u) atomicAdd(&sh.digit_base[(ksrc[s + i] >> shift) & 255u], 1u);
        __syncthreads();
        const uint32_t excl = ts_digit_scan(threadIdx.x < TS_RADIX ? sh.digit_base[threadIdx.x] : 0u, sh.wave_tot);
        if (threadIdx.x < TS_RADIX) sh.digit_base[threadIdx.x] = excl;
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += CHUNK) {
            const uint32_t m = min(CHUNK, n - c0);
            uint32_t key[TS_KMAX], val[TS_KMAX], pos[TS_KMAX];
#pragma unroll
            for (int r = 0; r < TS_KMAX; ++r) {
                const uint32_t i = first + r * 64u;
                key[r] = i < m ? ksrc[s + c0 + i] : TS_PAD;
                val[r] = i < m ? vsrc[s + c0 + i] : 0u;
            }
            digit_positions<W, TS_KMAX>(key, shift, sh, pos);
#pragma unroll
            for (int r = 0; r < TS_KMAX; ++r) {
                const uint32_t i = first + r * 64u;
                if (i < m) {  // (a padding key ranks behind the real keys of digit 255)
                    const uint32_t d = (key[r] >> shift) & 255u;
                    const uint32_t dst = s + sh.digit_base[d] + (pos[r] - sh.local_start[d]);
                    kdst[dst] = key[r];
                    vdst[dst] = val[r];
                }
            }
            __syncthreads();
            if (threadIdx.x < TS_RADIX) {  // advance the running offsets by this chunk's digit counts
                const uint32_t d = threadIdx.x;  // (padding sits at the tail of digit 255)
                const uint32_t end = d < 255u ? sh.local_start[d + 1] : m;
                sh.digit_base[d] += min(end, m) - min(sh.local_start[d], m);
            }
            __syncthreads();
        }
        __threadfence();  // pass 1 reads what pass 0 wrote through other lanes of this workgroup
        __syncthreads();
    }
}

// Small segments: one 256-lane workgroup per tile; segs[t] = [first, end) of the tile's pairs in the tile-sorted
// buffers (boundaries_kernel).  Tiles with more than 4096 pairs are only listed (big_list, big_count).
__global__ __launch_bounds__(TS_SMALL_WAVES * 64) void tile_sort_small_kernel(uint32_t *__restrict__ keys,
                                                                             uint32_t *__restrict__ vals,
                                                                             const uint2 *__restrict__ segs,
                                                                             uint32_t num_tiles,
                                                                             const uint32_t *__restrict__ d_count,
                                                                             uint32_t *__restrict__ big_count,
                                                                             uint32_t *__restrict__ big_list) {
    __shared__ TileSortShared<TS_SMALL_WAVES> sh;
    const uint32_t count = *d_count;
    const uint32_t t = blockIdx.x;
    const uint2 sg = segs[t];
    const uint32_t s = sg.x, e = min(sg.y, count);
    if (e <= s + 1u) return;  // empty or a single pair (workgroup-uniform)
    const uint32_t n = e - s;
    if (n <= 1024u) sort_segment_lds<TS_SMALL_WAVES, 4>(keys, vals, s, n, sh);
    else if (n <= 2048u) sort_segment_lds<TS_SMALL_WAVES, 8>(keys, vals, s, n, sh);
    else if (n <= 4096u) sort_segment_lds<TS_SMALL_WAVES, 16>(keys, vals, s, n, sh);
    else if (threadIdx.x == 0) big_list[atomicAdd(big_count, 1u)] = t;
}

// Long segments: 1024-lane workgroups walk the list; up to 16384 pairs stay in LDS, more are streamed through the
// other half of the ping-pong buffers.
__global__ __launch_bounds__(TS_BIG_WAVES * 64) void tile_sort_big_kernel(uint32_t *__restrict__ keys_a,
                                                                         uint32_t *__restrict__ vals_a,
                                                                         uint32_t *__restrict__ keys_b,
                                                                         uint32_t *__restrict__ vals_b,
                                                                         const uint2 *__restrict__ segs,
                                                                         const uint32_t *__restrict__ d_count,
                                                                         const uint32_t *__restrict__ big_count,
                                                                         const uint32_t *__restrict__ big_list) {
    extern __shared__ unsigned char ts_raw[];
    TileSortShared<TS_BIG_WAVES> &sh = *reinterpret_cast<TileSortShared<TS_BIG_WAVES> *>(ts_raw);
    const uint32_t count = *d_count, nb = *big_count;
    for (uint32_t k = blockIdx.x; k < nb; k += gridDim.x) {
        const uint2 sg = segs[big_list[k]];
        const uint32_t s = sg.x, n = min(sg.y, count) - sg.x;
        if (n <= 8192u) sort_segment_lds<TS_BIG_WAVES, 8>(keys_a, vals_a, s, n, sh);
        else if (n <= 16384u) sort_segment_lds<TS_BIG_WAVES, 16>(keys_a, vals_a, s, n, sh);
        else sort_segment_streamed<TS_BIG_WAVES>(keys_a, vals_a, keys_b, vals_b, s, n, sh);
    }
}

}  // namespace

// big_count: one device word, zeroed by the caller's frame set-up; big_list: num_tiles entries
int launch_tile_depth_sort(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, const uint2 *segs,
                           uint32_t num_tiles, const uint32_t *d_count, uint32_t *big_count, uint32_t *big_list,
                           hipStream_t s) {
    if (!num_tiles) return 0;
    // 146 KiB of dynamic LDS for the 1024-lane variant (per device and process state of the runtime: set every time,
    // this variant is opt-in and the call is cheap)
    const size_t big_lds = sizeof(TileSortShared<TS_BIG_WAVES>);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(tile_sort_big_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(tile_sort_small_kernel, dim3(num_tiles), dim3(TS_SMALL_WAVES * 64), 0, s, keys_a, vals_a, segs,
                       num_tiles, d_count, big_count, big_list);
    hipLaunchKernelGGL(tile_sort_big_kernel, dim3(TS_BIG_GRID), dim3(TS_BIG_WAVES * 64), big_lds, s, keys_a, vals_a,
                       keys_b, vals_b, segs, d_count, big_count, big_list);
    return 0;
}

}  // namespace gsplat
