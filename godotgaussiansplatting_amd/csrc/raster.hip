// Tile ranges + front-to-back tile compositor for gfx950 — replaces
// resources/shaders/compute/gsplat_boundaries.glsl and gsplat_render.glsl.
#include <type_traits>

#include "gsplat_internal.h"
#include "project_math.h"
#include "sh_eval.h"

namespace gsplat {

namespace {

// ---------------------------------------------------------------------------------------------------
// gsplat_boundaries.glsl:23-50.  Quirks kept for bit-exact tile_bounds: the highest populated tile
// never receives .y unless it is tile T-1, in which case .y = D-1.  Every lane whose tile is T-1 writes D-1 in the
// reference; the keys are sorted, so that is equivalent to the single lane i == D-1 writing it.
// D is read from device memory; a wave walks 256 consecutive keys per trip (one uint4 per lane).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void close_last(uint32_t *b, uint32_t cur, uint32_t i, uint32_t count, uint32_t num_tiles,
                                           int fix_last_tile, int sharded, const uint32_t *frame_last_tile_plus1) {
    // sharded frame: the quirk belongs to the whole frame's highest populated tile only
    const bool close_it = fix_last_tile || (sharded && cur + 1 != *frame_last_tile_plus1);
    if (close_it) {
        b[2 * cur + 1] = count;
    } else if (i > 0 && cur == num_tiles - 1) {
        b[2 * cur + 1] = count - 1;  // :47-49
    }
}

// KeyT = uint16_t: the sorted keys are tile ids (narrow-key frames, sort.hip)
// ... stripe-local ones (TileMap, gsplat_internal.h): equal / unequal is decided on the local ids, a tile's range is
// written at its global id (one integer division per CHANGE of tile, not per key)
template <typename KeyT>
__global__ __launch_bounds__(256) void boundaries_kernel(const KeyT *__restrict__ keys,
                                                         const uint32_t *__restrict__ d_count, uint32_t num_tiles,
                                                         uint2 *__restrict__ bounds, int fix_last_tile, int sharded,
                                                         const uint32_t *__restrict__ frame_last_tile_plus1,
                                                         uint32_t *__restrict__ last_tile_keep, TileMap map) {
    const uint32_t count = *d_count;
    uint32_t *b = reinterpret_cast<uint32_t *>(bounds);
    // (the word this pass asked with, kept for a replay of the frame: round 3 copied it with a 4-byte hipMemcpyAsync — a
    // blit kernel launch of its own in every frame)
    if (last_tile_keep != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *last_tile_keep = *frame_last_tile_plus1;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    for (uint64_t wbase = (uint64_t)wave_global * 256u; wbase < count; wbase += (uint64_t)waves * 256u) {
        const uint32_t i0 = (uint32_t)wbase + lane * 4u;
        uint32_t k[4];
        if (i0 + 3u < count) {
            if constexpr (sizeof(KeyT) == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(keys + i0);
                k[0] = v.x & 0xFFFFu; k[1] = v.x >> 16; k[2] = v.y & 0xFFFFu; k[3] = v.y >> 16;
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(keys + i0);
                k[0] = v.x; k[1] = v.y; k[2] = v.z; k[3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) k[e] = i0 + e < count ? keys[i0 + e] : 0u;
        }
        uint32_t prev = __shfl_up(k[3], 1, 64);
        if (lane == 0u) prev = i0 > 0u ? keys[i0 - 1u] : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = i0 + e;
            if (i < count) {
                constexpr bool LOCAL = sizeof(KeyT) == 2;
                constexpr int TILE_SHIFT = LOCAL ? 0 : 16;
                const uint32_t cur = k[e] >> TILE_SHIFT, pt = prev >> TILE_SHIFT;
                if (i > 0u && pt != cur) {
                    b[2 * (LOCAL ? map.global_of(pt) : pt) + 1] = i;   // .y
                    b[2 * (LOCAL ? map.global_of(cur) : cur) + 0] = i;  // .x
                }
                if (i == count - 1u)
                    close_last(b, LOCAL ? map.global_of(cur) : cur, i, count, num_tiles, fix_last_tile, sharded, frame_last_tile_plus1);
            }
            prev = k[e];
        }
    }
}

// Batched frames (gsplat_internal.h FrameBatch): the keys are stripe-local tile ids of the VIRTUAL frame, B real frames
// stacked; tile ranges are written at the virtual tile id, and quirk Q5/Q6 belongs to every REAL frame's highest populated
// tile: where the sorted array passes from frame f to a later one, the last tile of f is closed the way close_last closes the
// end of a single frame's array (which ends there) — with f's own "last tile + 1" word and f's real tile id.
__global__ __launch_bounds__(256) void boundaries_batch_kernel(const uint16_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ d_count,
                                                               uint2 *__restrict__ bounds, int fix_last_tile, int sharded,
                                                               const uint32_t *__restrict__ frame_last_tiles,
                                                               TileMap map, uint32_t rows, uint32_t real_sy0,
                                                               uint32_t real_tiles) {
    const uint32_t count = *d_count;
    uint32_t *b = reinterpret_cast<uint32_t *>(bounds);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t per_frame = rows * map.sw;  // local ids per frame
    // end of frame f's part of the array at index `end` (exclusive; its last key `lid` sits at end - 1)
    auto close_frame = [&](uint32_t lid, uint32_t end) {
        const uint32_t f = lid / per_frame, vrow = lid / map.sw, x = lid - vrow * map.sw;
        const uint32_t real = (real_sy0 + (vrow - f * rows)) * map.gx + map.sx0 + x;
        const uint32_t vglobal = map.global_of(lid);
        const bool close_it = fix_last_tile || (sharded && real + 1u != frame_last_tiles[f]);
        if (close_it) b[2 * vglobal + 1] = end;
        else if (end > 1u && real == real_tiles - 1u) b[2 * vglobal + 1] = end - 1u;  // gsplat_boundaries.glsl:47-49
    };
    for (uint64_t wbase = (uint64_t)wave_global * 256u; wbase < count; wbase += (uint64_t)waves * 256u) {
        const uint32_t i0 = (uint32_t)wbase + lane * 4u;
        uint32_t k[4];
        if (i0 + 3u < count) {
            const uint2 v = *reinterpret_cast<const uint2 *>(keys + i0);
            k[0] = v.x & 0xFFFFu; k[1] = v.x >> 16; k[2] = v.y & 0xFFFFu; k[3] = v.y >> 16;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) k[e] = i0 + e < count ? keys[i0 + e] : 0u;
        }
        uint32_t prev = __shfl_up(k[3], 1, 64);
        if (lane == 0u) prev = i0 > 0u ? keys[i0 - 1u] : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t i = i0 + e;
            if (i < count) {
                const uint32_t cur = k[e];
                if (i > 0u && prev != cur) {
                    b[2 * map.global_of(cur) + 0] = i;  // .x
                    if (prev / per_frame == cur / per_frame) b[2 * map.global_of(prev) + 1] = i;  // .y
                    else close_frame(prev, i);
                }
                if (i == count - 1u) close_frame(cur, count);
            }
            prev = k[e];
        }
    }
}

// Re-laid-out scene (gsplat_finalize_scene): the pairs were emitted in storage order, so the stable sort leaves equal
// keys in ascending STORAGE slot; the contract wants ascending splat id.  The same pass over the sorted keys repairs
// it into the other value buffer.  A wave looks at 64 consecutive sorted pairs:
//  * runs of equal keys inside the window are ranked with shuffles (one splat-id gather per tied element,
//    max-run-length rounds of ds_bpermute);
//  * a run that crosses a window edge but is at most 64 long is ranked by scanning it in memory (<= 64 x 2 loads);
//  * longer runs (a dense far field with one depth code; the zero-filled, not yet uploaded splats of a loading scene,
//    which all land on the origin's tile with one key) are only LISTED by their first element and sorted by
//    tie_long_kernel, one workgroup per run — O(L log L), never O(L^2).
constexpr uint32_t TIE_SHORT = 64;
__global__ __launch_bounds__(256) void boundaries_ties_kernel(const uint32_t *__restrict__ keys,
                                                              const uint32_t *__restrict__ d_count, uint32_t num_tiles,
                                                              uint2 *__restrict__ bounds, int fix_last_tile, int sharded,
                                                              const uint32_t *__restrict__ frame_last_tile_plus1,
                                                              uint32_t *__restrict__ last_tile_keep,
                                                              const uint32_t *__restrict__ tie_values_in,
                                                              uint32_t *__restrict__ tie_values_out,
                                                              const uint32_t *__restrict__ tie_id_of,
                                                              uint32_t *__restrict__ long_count,
                                                              uint32_t *__restrict__ long_list, uint32_t long_capacity) {
    const uint32_t count = *d_count;
    uint32_t *b = reinterpret_cast<uint32_t *>(bounds);
    if (last_tile_keep != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *last_tile_keep = *frame_last_tile_plus1;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long le_mask = lane == 63u ? ~0ull : ((2ull << lane) - 1ull);  // lanes <= me
    const unsigned long long ge_mask = ~0ull << lane;                                  // lanes >= me
    for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < count; base += gridDim.x * blockDim.x) {
        const uint32_t i = base + lane;
        const bool valid = i < count;
        const uint32_t key = valid ? keys[i] : 0u;
        const uint32_t cur = key >> 16;
        // neighbours' keys: inside the wave by shuffle, across the window edge from memory
        uint32_t prev_key = __shfl_up(key, 1, 64);
        if (lane == 0u) prev_key = (valid && i > 0) ? keys[i - 1] : ~key;
        const bool has_prev = valid && i > 0;
        if (has_prev) {
            const uint32_t prev = prev_key >> 16;
            if (prev != cur) {
                b[2 * prev + 1] = i;  // .y
                b[2 * cur + 0] = i;   // .x
            }
        }
        if (valid && i == count - 1) close_last(b, cur, i, count, num_tiles, fix_last_tile, sharded, frame_last_tile_plus1);

        const uint32_t v = valid ? tie_values_in[i] : 0u;
        uint32_t next_key = __shfl_down(key, 1, 64);
        const bool has_next = valid && (i + 1 < count);
        if (lane == 63u) next_key = has_next ? keys[i + 1] : ~key;
        const bool tie_prev = has_prev && prev_key == key;
        const bool tie_next = has_next && next_key == key;
        const bool in_tie = tie_prev || tie_next;
        const unsigned long long heads = __ballot(valid && !tie_prev);   // first element of a run (or a singleton)
        const unsigned long long tails = __ballot(valid && !tie_next);   // last element of a run
        const unsigned long long h_le = heads & le_mask, t_ge = tails & ge_mask;
        const bool closed = in_tie && h_le != 0ull && t_ge != 0ull;     // the whole run lies in this window
        // a run that leaves the window: how far does it go?  (<= TIE_SHORT loads each way)
        uint32_t s0 = i, e0 = i + 1;
        bool is_long = false;
        if (in_tie && !closed) {
            while (s0 > 0 && i - s0 < TIE_SHORT && keys[s0 - 1] == key) --s0;
            if (i - s0 >= TIE_SHORT) is_long = true;
            while (!is_long && e0 < count && keys[e0] == key) {
                ++e0;
                if (e0 - s0 > TIE_SHORT) is_long = true;
            }
        }
        const uint32_t my_id = (in_tie && !is_long) ? tie_id_of[v] : 0u;
        const uint32_t s_lane = h_le ? 63u - (uint32_t)__builtin_clzll(h_le) : 0u;
        const uint32_t e_lane = t_ge ? (uint32_t)__builtin_ctzll(t_ge) : 63u;
        uint32_t len = closed ? e_lane - s_lane + 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, d, 64));
        uint32_t rank = 0;
        for (uint32_t k = 0; k < len; ++k) {  // len = longest closed run of the window (wave-uniform)
            const uint32_t src = s_lane + k;
            const uint32_t other = __shfl(my_id, (int)(src & 63u), 64);
            if (closed && src <= e_lane) rank += other < my_id ? 1u : 0u;
        }
        uint32_t dst = i;
        bool write = valid;
        if (closed) {
            dst = base + s_lane + rank;
        } else if (in_tie && !is_long) {
            uint32_t r = 0;
            for (uint32_t j = s0; j < e0; ++j) r += tie_id_of[tie_values_in[j]] < my_id ? 1u : 0u;
            dst = s0 + r;
        } else if (is_long) {
            write = false;  // tie_long_kernel writes the whole run
            if (!tie_prev) {
                const uint32_t slot = atomicAdd(long_count, 1u);
                if (slot < long_capacity) long_list[slot] = i;
            }
        }
        if (write) tie_values_out[dst] = v;
    }
}

// One 1024-lane workgroup per listed run [s0, e0) of equal keys: order its storage slots by splat id.
// Up to 4096 elements: bitonic sort of (id, slot) in LDS.  Longer: workgroup-serial LSD radix sort on the id bits
// through global scratch — the run's own ranges of the spare key buffer / the output (A) and of the sorted keys / the
// input values (B); every key of the run is the same word, so the sorted keys are restored from it afterwards.
constexpr uint32_t TIE_LDS_MAX = 4096;
__global__ __launch_bounds__(1024) void tie_long_kernel(uint32_t *__restrict__ keys_sorted,
                                                        uint32_t *__restrict__ keys_scratch,
                                                        uint32_t *__restrict__ values_in,
                                                        uint32_t *__restrict__ values_out,
                                                        const uint32_t *__restrict__ d_count,
                                                        const uint32_t *__restrict__ id_of, uint32_t n_splats,
                                                        const uint32_t *__restrict__ long_count,
                                                        const uint32_t *__restrict__ long_list, uint32_t long_capacity) {
    __shared__ uint32_t s_a[TIE_LDS_MAX];   // bitonic: ids        | radix: per-wave digit counts [16][256]
    __shared__ uint32_t s_b[TIE_LDS_MAX];   // bitonic: slots      | radix: [0,256) histogram / running digit offsets
    __shared__ uint32_t s_end;
    const uint32_t count = *d_count;
    const uint32_t runs = min(*long_count, long_capacity);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t e = blockIdx.x; e < runs; e += gridDim.x) {
        const uint32_t s0 = long_list[e];
        const uint32_t key = keys_sorted[s0];
        if (tid == 0) s_end = 0xFFFFFFFFu;
        __syncthreads();
        for (uint32_t pos = s0 + 1;; pos += 1024u) {  // first index past the run
            const uint32_t idx = pos + tid;
            if (idx >= count || keys_sorted[idx] != key) atomicMin(&s_end, min(idx, count));
            __syncthreads();
            if (s_end != 0xFFFFFFFFu) break;
            __syncthreads();
        }
        const uint32_t e0 = s_end, len = e0 - s0;
        __syncthreads();
        if (len <= TIE_LDS_MAX) {
            uint32_t np = 64;
            while (np < len) np <<= 1;
            for (uint32_t j = tid; j < np; j += 1024u) {
                const uint32_t slot = j < len ? values_in[s0 + j] : 0u;
                s_b[j] = slot;
                s_a[j] = j < len ? id_of[slot] : 0xFFFFFFFFu;
            }
            __syncthreads();
            for (uint32_t k = 2; k <= np; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t t = tid; t < np / 2; t += 1024u) {
                        const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                        const bool up = (lo & k) == 0;
                        const uint32_t a = s_a[lo], c = s_a[hi];
                        if ((a > c) == up) {
                            s_a[lo] = c; s_a[hi] = a;
                            const uint32_t sa = s_b[lo];
                            s_b[lo] = s_b[hi]; s_b[hi] = sa;
                        }
                    }
                    __syncthreads();
                }
            for (uint32_t j = tid; j < len; j += 1024u) values_out[s0 + j] = s_b[j];
            __syncthreads();
            continue;
        }
        // ---- long run: LSD radix on the id bits, ping-pong A <-> B, ending in A (values_out)
        uint32_t bits = 1;
        while (bits < 32 && (1ull << bits) < (unsigned long long)n_splats) ++bits;
        const uint32_t passes = (bits + 7u) / 8u;
        uint32_t *ids[2] = {keys_scratch + s0, keys_sorted + s0};  // [0] = A, [1] = B
        uint32_t *slots[2] = {values_out + s0, values_in + s0};
        uint32_t cur = passes & 1u;  // odd number of passes: start in B
        for (uint32_t j = tid; j < len; j += 1024u) {
            const uint32_t slot = values_in[s0 + j];
            ids[cur][j] = id_of[slot];
            if (cur == 0u) slots[0][j] = slot;
        }
        __syncthreads();
        uint32_t(*wcnt)[256] = reinterpret_cast<uint32_t(*)[256]>(s_a);
        uint32_t *offs = s_b;
        for (uint32_t pass = 0; pass < passes; ++pass) {
            const uint32_t shift = 8u * pass;
            const uint32_t *src_id = ids[cur], *src_slot = slots[cur];
            uint32_t *dst_id = ids[cur ^ 1u], *dst_slot = slots[cur ^ 1u];
            if (tid < 256u) offs[tid] = 0u;
            __syncthreads();
            for (uint32_t j = tid; j < len; j += 1024u) atomicAdd(&offs[(src_id[j] >> shift) & 255u], 1u);
            __syncthreads();
            if (wave == 0) {  // exclusive scan of the 256 digit counts: 4 per lane
                uint32_t c[4], sum = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { c[q] = offs[lane * 4u + q]; sum += c[q]; }
                uint32_t incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t t = __shfl_up(incl, d, 64);
                    if ((int)lane >= d) incl += t;
                }
                uint32_t run = incl - sum;
#pragma unroll
                for (int q = 0; q < 4; ++q) { offs[lane * 4u + q] = run; run += c[q]; }
            }
            __syncthreads();
            for (uint32_t chunk = 0; chunk < len; chunk += 1024u) {
                for (uint32_t q = tid; q < 16u * 256u; q += 1024u) s_a[q] = 0u;
                __syncthreads();
                const uint32_t j = chunk + tid;
                const bool ok = j < len;
                const uint32_t idv = ok ? src_id[j] : 0u, slv = ok ? src_slot[j] : 0u;
                const uint32_t d = (idv >> shift) & 255u;
                unsigned long long m = __ballot(ok);
#pragma unroll
                for (int bq = 0; bq < 8; ++bq) {
                    const bool bit = (d >> bq) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    m &= bit ? bal : ~bal;
                }
                const uint32_t in_group = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (ok && (m >> lane) <= 1ull) wcnt[wave][d] = in_group + 1u;  // highest lane of the group: its size
                __syncthreads();
                if (tid < 256u) {  // digit = tid: wave-exclusive prefixes on top of the running offset
                    uint32_t run = offs[tid];
                    for (int w = 0; w < 16; ++w) {
                        const uint32_t c = wcnt[w][tid];
                        wcnt[w][tid] = run;
                        run += c;
                    }
                    offs[tid] = run;
                }
                __syncthreads();
                if (ok) {
                    const uint32_t dst = wcnt[wave][d] + in_group;
                    dst_id[dst] = idv;
                    dst_slot[dst] = slv;
                }
                __syncthreads();
            }
            cur ^= 1u;
        }
        for (uint32_t j = tid; j < len; j += 1024u) keys_sorted[s0 + j] = key;  // B's id scratch was the sorted keys
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// gsplat_render.glsl:50-111.  One 256-lane workgroup (4 wave64) per 16x16 tile, one lane per pixel.
// Splats are staged through LDS in batches of 256 (36 B each: centre, the conic pre-multiplied into
// (hx,hy,hz) = (-0.5cx, -cy, -0.5cz)*log2(e), opacity, rgb) and broadcast-read by every lane.
// Early termination follows the reference exactly: a lane stops when t <= 1/255; the
// workgroup fetches the next batch only while sum_lanes uint(t*255) > 255, where out-of-image lanes of
// edge tiles take part in the sum.  The sum is a wave reduction + one LDS atomic per wave.
// exp() follows the arithmetic contract; FAST_EXP swaps in the hardware v_exp_f32.
// ---------------------------------------------------------------------------------------------------
constexpr float LOG2E = 0x1.715476p+0f;
constexpr float MIN_ALPHA = 1.0f / 255.0f;  // gsplat_render.glsl:7 (0x3b808081 in blend_list)
// Contract: exp(power) is exactly 0 when power*log2(e) < -32 (< 2.4e-10): the splat leaves
// that pixel's colour and transmittance untouched.  A wave none of whose live pixels is above the cutoff skips the
// splat after 8 of its ~26 VALU instructions (~37 % of the wave-steps at 6 M splats, 1080p).
constexpr float EXP_CUTOFF = -32.0f;  // (0xc2000000 in blend_list)
static_assert(MIN_ALPHA == 0x1.010102p-8f && EXP_CUTOFF == -0x1p+5f, "blend_list carries these as literals");
static_assert(LOG2E == STAGE_LOG2E, "project_math.h: staged_geometry pre-multiplies the conic with the same constant");

// 2^y per the contract: y clamped to [-125, 126], n = rint(y) (round-half-even), f = y - n, degree-5 polynomial
// p(f) with p(0) = 1, result p * 2^n.  Evaluated here without cvt/ldexp: adding 1.5*2^23 leaves n in the low
// mantissa bits of `big` (same rounding as rint), and because p is in [0.70, 1.42] and n >= -125 the product is a
// normal number, so p * 2^n is an integer add of n to p's exponent field — bit-identical to ldexpf(p, n).
template <bool FAST_EXP>
__device__ __forceinline__ float exp2_contract(float y) {
    if (FAST_EXP) return __builtin_amdgcn_exp2f(y);
    // (the lower clamp of the contract, -125, is left out: the compositor only uses the result where y >= -32; one
    // v_min_f32 — fminf() would add a canonicalising v_max in IEEE mode, fmed3 a register for its second constant)
    asm("v_min_f32 %0, 0x42fc0000, %1" : "=v"(y) : "v"(y));  // min(126.0f, y)
    const float big = y + 12582912.0f;
    const float n = big - 12582912.0f;
    const float f = y - n;
    float q = __builtin_fmaf(0x1.5bba18p-10f, f, 0x1.3cea88p-7f);
    q = __builtin_fmaf(q, f, 0x1.c6b752p-5f);
    q = __builtin_fmaf(q, f, 0x1.ebf9bcp-3f);
    q = __builtin_fmaf(q, f, 0x1.62e42ap-1f);
    const float p = __builtin_fmaf(q, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (__float_as_uint(big) << 23));
}

// LDS byte address of a __shared__ object (what ds_read takes)
__device__ __forceinline__ uint32_t lds_address(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

// The blend loop of one wave over its work list (gsplat_render.glsl:79-91), written out instruction by instruction.
// list = LDS address of `count` u32 entries (each the LDS address of a staged 48-byte record {ipx, ipy, hx, hy}{hz, -,
// -, -}{r, g, b, opacity}) followed by two readable entries; px, py = the lane's pixel; t, cr, cg, cb = its
// state.  The arithmetic is the contract's, operation for operation what the C++ form of round 2 compiled to
// (exponent: dx = ipx - px, dy = ipy - py, a1 = fma(hy, dy, hx dx), y = fma(a1, dx, (hz dy) dy); exp2_contract;
// alpha = opacity e; w = alpha t; c = fma(rgb, w, c); t = t - w) — what is hand-written is everything around it:
//  * exec IS the set of alive pixels for the whole loop (v_cmpx retires a pixel when t <= 1/255, the wave leaves on
//    execz), so v_cmp of y against the cutoff under it yields "alive and sees the splat" directly: s_cbranch_vccz
//    skips the splat for the wave, s_and_saveexec narrows exec to the lanes that take the update.  hipcc's form of
//    the same logic spent a second v_cmp + a v_cndmask on vcc per step (alpha = above ? .. : 0) — 4 of the step's 61.5
//    cycles (tools/step_rates.hip; back to back such a v_cndmask issues every 20 cycles) — and nine scalar instructions.
//  * three v_fmac for the colour instead of the v_pk_fma + v_fmac the SLP vectoriser makes of them (a packed f32 FMA
//    issues slower than two scalar ones).
//  * software-pipelined: the geometry of entry i+1 and list entry i+2 are requested at the start of step i (below).
// 25.2 VALU per seen step (27.5 before), 8 per step that ends at the cutoff test.  Measured per step per SIMD at 8
// waves, records in registers: 54.7 vs 61.7 cycles (profiles/r03_step_rates.md); c3 launch 0.330 -> 0.305 ms, with
// the pipelining 0.292 (the same pipelining bought nothing in round 3's C++ loop, which was VALU-bound outright).
// (The upper clamp of exp2_contract is worth 2 % of the launch and stays: y <= 126 holds for every record a sane
// scene produces, but the bound grows with the square of the image diagonal and the contract has no size limit.)
// FAST_EXP swaps the polynomial for v_exp_f32 (opt-in build, not the contract).
//
// 2^y of v11 into v10 (exp2_contract above, same operations in the same order)
#define GS_EXP2_HEAD                                                                                                \
    "v_min_f32_e32 v12, 0x42fc0000, v11\n"                                                                          \
    "v_add_f32_e32 v13, 0x4b400000, v12\n"                                                                          \
    "v_add_f32_e32 v10, 0xcb400000, v13\n"                                                                          \
    "v_sub_f32_e32 v12, v12, v10\n"
#define GS_EXP2_CONTRACT                                                                                            \
    GS_EXP2_HEAD                                                                                                    \
    "v_fmamk_f32 v10, v12, 0x3aaddd0c, %[c4]\n"                                                                     \
    "v_fmaak_f32 v10, v10, v12, 0x3d635ba9\n"                                                                       \
    "v_fmaak_f32 v10, v10, v12, 0x3e75fcde\n"                                                                       \
    "v_fmaak_f32 v10, v10, v12, 0x3f317215\n"                                                                       \
    "v_fma_f32 v10, v10, v12, 1.0\n"                                                                                \
    "v_lshl_add_u32 v10, v13, 23, v10\n"
#define GS_EXP2_HARDWARE "v_exp_f32_e32 v10, v11\n s_nop 0\n"
// One step.  The geometry of entry i+1 and the list entry i+2 are requested at the start of step i, so a step's
// exponent never waits for LDS (the requests of the previous step have had a whole step to land).  Entries
// rotate through three registers (entry i is still the colour read's address while i+1 and i+2 are in use), geometry
// through two sets: six steps per trip.  The list needs two readable entries behind its end.
#define GS_PSTEP(...) GS_PSTEP_(__VA_ARGS__)
#define GS_PSTEP_(EC, EN, EL, OFFL, G0, G1, G2, G3, GC, HC, GN, HN, EXP2)                                           \
    "s_waitcnt lgkmcnt(0)\n"                                                                                        \
    "ds_read_b128 " GN ", " EN "\n"                                                                                 \
    "ds_read_b32 " HN ", " EN " offset:16\n"                                                                        \
    "ds_read_b32 " EL ", %[la] offset:" OFFL "\n"                                                                   \
    "v_sub_f32_e32 v8, " G0 ", %[px]\n"                                                                             \
    "v_sub_f32_e32 v9, " G1 ", %[py]\n"                                                                             \
    "v_mul_f32_e32 v10, " G2 ", v8\n"                                                                               \
    "v_mul_f32_e32 v11, " HC ", v9\n"                                                                               \
    "v_fmac_f32_e32 v10, " G3 ", v9\n"                                                                              \
    "v_mul_f32_e32 v11, v11, v9\n"                                                                                  \
    "v_fmac_f32_e32 v11, v10, v8\n"                                                                                 \
    "v_cmp_le_f32_e32 vcc, 0xc2000000, v11\n"                                                                       \
    "s_cbranch_vccz 2f\n"                                                                                           \
    "s_and_saveexec_b64 %[tm], vcc\n"                                                                               \
    "ds_read_b128 " GC ", " EC " offset:32\n"                                                                       \
    EXP2                                                                                                            \
    "s_waitcnt lgkmcnt(0)\n"                                                                                        \
    "v_mul_f32_e32 v10, " G3 ", v10\n"                                                                              \
    "v_mul_f32_e32 v10, v10, %[t]\n"                                                                                \
    "v_fmac_f32_e32 %[cr], " G0 ", v10\n"                                                                           \
    "v_fmac_f32_e32 %[cg], " G1 ", v10\n"                                                                           \
    "v_fmac_f32_e32 %[cb], " G2 ", v10\n"                                                                           \
    "v_sub_f32_e32 %[t], %[t], v10\n"                                                                               \
    "s_mov_b64 exec, %[tm]\n"                                                                                       \
    "v_cmpx_lt_f32_e32 0x3b808081, %[t]\n"                                                                          \
    "s_cbranch_execz 9f\n"                                                                                          \
    "2:\n"                                                                                                          \
    "s_sub_u32 %[n], %[n], 1\n"                                                                                     \
    "s_cbranch_scc1 9f\n"
#define GS_SET_A "v4", "v5", "v6", "v7", "v[4:7]", "v3"
#define GS_SET_B "v14", "v15", "v16", "v17", "v[14:17]", "v18"
#define GS_BLEND_LOOP(EXP2)                                                                                         \
    "s_mov_b64 %[sv], exec\n"                                                                                       \
    "v_cmpx_lt_f32_e32 0x3b808081, %[t]\n"                                                                          \
    "s_cbranch_execz 9f\n"                                                                                          \
    "ds_read_b32 v0, %[la]\n"                                                                                       \
    "ds_read_b32 v1, %[la] offset:4\n"                                                                              \
    "s_waitcnt lgkmcnt(1)\n"                                                                                        \
    "ds_read_b128 v[4:7], v0\n"                                                                                     \
    "ds_read_b32 v3, v0 offset:16\n"                                                                                \
    "1:\n"                                                                                                          \
    GS_PSTEP("v0", "v1", "v2", "8", GS_SET_A, "v[14:17]", "v18", EXP2)                                              \
    GS_PSTEP("v1", "v2", "v0", "12", GS_SET_B, "v[4:7]", "v3", EXP2)                                                \
    GS_PSTEP("v2", "v0", "v1", "16", GS_SET_A, "v[14:17]", "v18", EXP2)                                             \
    GS_PSTEP("v0", "v1", "v2", "20", GS_SET_B, "v[4:7]", "v3", EXP2)                                                \
    GS_PSTEP("v1", "v2", "v0", "24", GS_SET_A, "v[14:17]", "v18", EXP2)                                             \
    GS_PSTEP("v2", "v0", "v1", "28", GS_SET_B, "v[4:7]", "v3", EXP2)                                                \
    "v_add_u32_e32 %[la], 24, %[la]\n"                                                                              \
    "s_branch 1b\n"                                                                                                 \
    "9:\n"                                                                                                          \
    "s_mov_b64 exec, %[sv]\n"                                                                                       \
    "s_waitcnt lgkmcnt(0)\n"
#define GS_CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "vcc", "scc", "memory"
template <bool FAST_EXP>
__device__ __forceinline__ void blend_list(uint32_t list, int count, float px, float py, float &t, float &cr, float &cg, float &cb) {
    unsigned long long sv, tm;
    int n = count - 1;
    const float c4 = 0x1.3cea88p-7f;
    if (FAST_EXP)
        asm volatile(GS_BLEND_LOOP(GS_EXP2_HARDWARE)
                     : [t] "+v"(t), [cr] "+v"(cr), [cg] "+v"(cg), [cb] "+v"(cb), [la] "+v"(list), [n] "+s"(n), [sv] "=&s"(sv), [tm] "=&s"(tm)
                     : [px] "v"(px), [py] "v"(py), [c4] "v"(c4)
                     : GS_CLOBBERS);
    else
        asm volatile(GS_BLEND_LOOP(GS_EXP2_CONTRACT)
                     : [t] "+v"(t), [cr] "+v"(cr), [cg] "+v"(cg), [cb] "+v"(cb), [la] "+v"(list), [n] "+s"(n), [sv] "=&s"(sv), [tm] "=&s"(tm)
                     : [px] "v"(px), [py] "v"(py), [c4] "v"(c4)
                     : GS_CLOBBERS);
}

// Conservative reach test of one staged splat against the four 8x8 pixel quadrants of its tile (bit w = wave w).
// The compositor ignores a splat for a pixel when y = power*log2(e) < -32 as evaluated in f32 (EXP_CUTOFF); with
// A = -hx, B = -hy, C = -hz the region y >= -K is the ellipse Q(dx, dy) = A dx^2 + B dx dy + C dy^2 <= K.  A quadrant
// can only hold such a pixel if the minimum of Q over its rectangle of pixel centres [x, x+7] x [y, y+7] is <= K (the
// rectangle contains every pixel).  Q is a convex quadratic with its minimum 0 at the splat's centre: if the centre is
// inside the rectangle the minimum is 0; otherwise it lies on an edge FACING the centre — the vertical edge at the
// rectangle's nearest dx (if the centre is outside the x range) and/or the horizontal one at its nearest dy — and on
// such an edge at the stationary point -B c / (2C) (resp. -B c / (2A)) clamped to the edge.  (Round 2 tested the
// ellipse's bounding box only, i.e. the two unclamped edge minima: 25 % of all wave-steps at 6 M splats / 1080p went
// away, but of the rest another 37 % still ended at the cutoff test in the blend loop — quadrants diagonal to an
// elongated splat.  Those are what the clamping is after.)
// K = 34 against the cutoff's 32: the f32 evaluation error of y in the blend loop is below 6 eps * (|hx|dx^2 +
// |hy dx dy| + |hz|dy^2) <= 12 eps * Q / (1 - rho), rho = |B| / (2 sqrt(AC)), and the test is only used when
// 1 - rho > 1e-4 (4AC - B^2 > 2.5e-4 * 4AC), which bounds that error by 0.01 Q and the cancellation inside the edge
// minima by 4e-4 relative — the 6 % between 32 and 34 covers both.  Indefinite, degenerate or NaN conics reach every
// quadrant (all comparisons false).  So skipping on this mask is invisible in the output: it only removes wave-steps
// that the blend loop would have rejected after evaluating y.  Measured at 6 M splats / 1080p: 40 % of the steps that
// used to end at the cutoff test are gone (VALU instructions of the launch -4 %, time -2 %); what is left of them is
// 10 % of the kernel's VALU work.  (Shrinking the rectangle to the bounding box of the quadrant's ALIVE pixels — a
// pixel that has left the loop never returns — removed nothing measurable: built, bit-exact, r3e in DESIGN.md §7.)
__device__ __forceinline__ uint32_t quadrant_mask(float sx, float sy, float A, float B, float C, float ox, float oy) {
    const float ac4 = (4.0f * A) * C;
    const float det4 = ac4 - B * B;
    if (!(A > 0.0f && C > 0.0f && det4 > 2.5e-4f * ac4)) return 0xFu;
    constexpr float K = 34.0f;
    const float sty = -B / (2.0f * C), stx = -B / (2.0f * A);  // stationary dy for a given dx, and vice versa
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float qx = ox + (float)((q & 1) * 8), qy = oy + (float)((q >> 1) * 8);
        const float dx_hi = sx - qx, dx_lo = dx_hi - 7.0f, dy_hi = sy - qy, dy_lo = dy_hi - 7.0f;
        // the rectangle's point nearest to the centre, per axis (0 where the centre is inside the range)
        const float cx = __builtin_amdgcn_fmed3f(0.0f, dx_lo, dx_hi), cy = __builtin_amdgcn_fmed3f(0.0f, dy_lo, dy_hi);
        const float ty = __builtin_amdgcn_fmed3f(sty * cx, dy_lo, dy_hi);   // on the edge dx = cx
        const float tx = __builtin_amdgcn_fmed3f(stx * cy, dx_lo, dx_hi);   // on the edge dy = cy
        const float m1 = (A * cx) * cx + ((B * cx) + C * ty) * ty;
        const float m2 = (C * cy) * cy + ((B * cy) + A * tx) * tx;
        // (an axis whose range holds the centre contributes no facing edge; both: the centre is inside, minimum 0)
        const float m = cx != 0.0f ? (cy != 0.0f ? fminf(m1, m2) : m1) : (cy != 0.0f ? m2 : 0.0f);
        mask |= (m > K) ? 0u : (1u << q);  // (NaN: kept)
    }
    return mask;
}

// 8 waves per SIMD (<= 64 VGPRs): the staging code of the lazy variants would take more and halve the occupancy of
// the VALU-bound inner loop
#ifndef GSPLAT_RENDER_MINWAVES
#define GSPLAT_RENDER_MINWAVES 8
#endif
// DEG >= 1: "lazy" frame — RasterizeData holds no colours; get_color (gsplat_projection.glsl:94-121,198-201) is
// evaluated here, with bands 0..DEG, for every splat the tile stages, i.e. only for splats that are composited (at
// 6 M splats / 1080p the block early exit leaves half of the visible splats uncomposited).  Same expression as the
// projection kernel's (sh_eval.h), so the image cannot tell who evaluated a colour.  DEG == 0: the colours are final.
// ROUND 0: the whole frame in one launch.  1 / 2: the two rounds of a two-round frame (projection.hip): round 1
// composites the front of the depth-sorted list and leaves, for every tile it did not finish, the per-pixel state
// (r, g, b, transmittance) in the image and the number of pairs consumed in tile_staged; round 2 picks unfinished tiles
// up from there, with the remaining pairs and the reference's batch boundaries (multiples of 256 pairs of the tile's
// WHOLE list).  The frame's last tile T - 1 loses the last pair of its COMPLETE list (quirk Q6), which round 1 cannot
// know: round 1 composites it without the last pair it has (its tile range already ends one short), and if the tile
// saturates on that — the reference then never gets to the dropped pair either — the result stands; if not, it is thrown
// away and round 2 composites T - 1 from scratch, from the complete list.
// GEO (lazy frames only, DEG >= 1): `culled` holds the 32-byte STAGED geometry of every visible splat ({ipx, ipy, hx, hy}
// {hz, opacity, -, -}: project_math.h staged_geometry, written by this frame's projection kernel, which has computed all
// of it for the tile rectangle anyway) — the staging branch gathers those 32 bytes and evaluates only get_color from the
// splat's scene slot, instead of recomputing the whole projection of every pair it stages (~600 VALU instructions and
// eleven correctly rounded divisions per staged pair, in every tile that stages the splat) inside the one kernel of the
// frame that is bound by VALU issue.  Same bits: one expression per quantity, whoever evaluates it.
// BATCH (batched frames, gsplat_internal.h FrameBatch; ROUND 0 only): `fp` is the VIRTUAL frame — the schedule, the tile
// ranges and the staged counts are indexed by virtual tile (B real stripes stacked) — and a tile's frame supplies what is
// real about it: its pixels (image + frame * stride, real tile row), its camera (colours of a lazy frame) and the scene
// slot behind a listed value (value = frame * n_pad + slot).
template <bool FAST_EXP, int DEG, int ROUND, bool GEO = false, bool BATCH = false>
__global__ __launch_bounds__(256, GSPLAT_RENDER_MINWAVES) void render_kernel(const float4 *__restrict__ culled,
                                                     const float4 *__restrict__ sh_block,
                                                     const uint32_t *__restrict__ values,
                                                     const uint2 *__restrict__ bounds, FrameParams fp,
                                                     float4 *__restrict__ image, uint32_t pitch_px, uint32_t origin_x,
                                                     uint32_t origin_y, float4 *__restrict__ pick,
                                                     uint32_t *__restrict__ tile_staged,
                                                     const uint32_t *__restrict__ tile_order,
                                                     uint32_t *__restrict__ tile_done,
                                                     FramePlan *__restrict__ plan,
                                                     float *__restrict__ edge_t,
                                                     std::conditional_t<BATCH, FrameBatch, NoBatch> batch) {
    static_assert(!BATCH || ROUND == 0, "batched frames are composited in one round");
    // one 48-byte record per staged splat: {ipx, ipy, hx, hy} {hz, -, -, -} {r, g, b, opacity}; all lanes of a wave
    // read the same record (LDS broadcast): one b128 + one b32 for the geometry, one b128 for colour and opacity
    // — every read naturally aligned (an LDS read is billed per lane whatever it broadcasts: b64 / b128 move 8 bytes per
    // cycle and lane, b32 four, unaligned pieces less: MI355X_MICROARCH.md §LDS)
    __shared__ float4 s_rec[256 * 3];
    __shared__ uint32_t s_sum;
    // quadrant prefilter: s_mask[j] bit w = staged splat j can reach wave w's 8x8 quadrant; s_list[w] = the byte
    // offsets (into s_rec) of the splats wave w has to look at, in list order
    __shared__ uint8_t s_mask[256];
    __shared__ __attribute__((aligned(8))) uint32_t s_list[4][256 + 2];  // (rows of 1032 bytes: read two entries at a time)

    // Tile schedule.  With tile_order (scan_blocks_kernel: the stripe's tiles, most expensive first by the previous
    // frame's staged count) workgroup b takes tile_order[b]: the dispatcher hands out workgroups in index order, so the
    // long tiles start first and the launch drains on short ones.  The table is eight interleaved lists (ORDER_XCD,
    // gsplat_internal.h): workgroup b runs on XCD b % 8 (observed, not a contract — only speed depends on it), and slot
    // b holds a tile of XCD b % 8's own blocks, so tiles that gather the same records meet in one L2; empty slots
    // (partial blocks) are ~0.  Without a table (the one-tile pick launch, stripes too large for the one-workgroup
    // sort): tile row r of the stripe goes to XCD r % 8.
    uint32_t bx, by;
    if (tile_order != nullptr) {
        const uint32_t t = tile_order[blockIdx.x];
        if (t == ~0u) return;
        bx = t % fp.gx; by = t / fp.gx;
    } else {
        const uint32_t stripe_w = fp.sx1 - fp.sx0, stripe_h = fp.sy1 - fp.sy0;
        const uint32_t slot = blockIdx.x >> 3;
        const uint32_t row = (slot / stripe_w) * 8u + (blockIdx.x & 7u);
        if (row >= stripe_h) return;
        bx = fp.sx0 + slot % stripe_w; by = fp.sy0 + row;
    }
    const uint32_t tile_id = by * fp.gx + bx;   // (BATCH: the virtual tile — what bounds / tile_staged are indexed by)
    uint32_t frame = 0, id_base = 0;
    if constexpr (BATCH) {  // the real tile row, the frame's images and cameras
        frame = by / batch.rows;
        by = batch.f[frame].sy0 + (by - frame * batch.rows);
        id_base = frame * batch.n_pad;
        image += (size_t)frame * batch.image_stride_px;
    }
    const FrameParams &rf = frame_of<BATCH>(fp, batch, frame);
    const uint32_t tid = threadIdx.y * TILE + threadIdx.x;
    const int lane = tid & 63;
#ifdef GS_PROBE_TIMELINE  // tools/render_timeline.py: where a wave's time goes (written over two pixels of its quadrant)
    const unsigned long long probe_t0 = __builtin_readcyclecounter();
    uint32_t probe_blend = 0, probe_steps = 0, probe_stage = 0, probe_bar2 = 0, probe_list = 0, probe_tail = 0;
#endif
    // wave w owns the 8x8 pixel quadrant (w&1, w>>1) of the tile: a compact footprint, so more splats are out of
    // reach of a whole wave (cutoff skip) than with 16x4 strips
    const uint32_t loc_x = ((tid >> 6) & 1u) * 8u + (lane & 7u), loc_y = (tid >> 7) * 8u + ((uint32_t)lane >> 3);
    const uint32_t pix_x = bx * TILE + loc_x, pix_y = by * TILE + loc_y;
    const float pxf = (float)pix_x, pyf = (float)pix_y;  // :58 integer pixel centres

    // two-round frames: which tiles this launch composites, and from which state
    int consumed = 0;          // pairs of the tile's whole list composited before this launch
    bool resume = false;
    bool whole_frame = ROUND == 0;  // this launch sees the tile's complete list
    if constexpr (ROUND == 1) {
        whole_frame = plan->single != 0u;
    }
    if constexpr (ROUND == 2) {
        if (plan->single != 0u) return;
        if (tile_done[tile_id] != 0u) return;  // finished (and written) by round 1
        if (tile_id != fp.gx * fp.gy - 1u) {   // (T - 1: from scratch, from its complete list)
            consumed = (int)tile_staged[tile_id];
            resume = true;
        }
    }

    const uint2 bnd = bounds[tile_id];
    int num = (int)(bnd.y - bnd.x);  // :61
    num = num < 0 ? 0 : num;

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, t = 1.0f;
    // Out-of-image lanes of an edge tile (W or H not a multiple of 16) take part in the block early-exit sum (:66,97,
    // SURVEY Q7), so their transmittance is state of the tile like any pixel's: between the rounds it waits in
    // edge_t[edge tile][lane] (edge tiles: the bottom row, then the right column)
    const bool in_image = pix_x < rf.width && pix_y < rf.height;
    const uint32_t edge_slot = (by == fp.gy - 1u ? bx : fp.gx + by) * 256u + tid;
    if (ROUND == 2 && resume) {
        if (in_image) {
            const float4 st = image[(size_t)(pix_y - origin_y) * pitch_px + (pix_x - origin_x)];
            cr = st.x; cg = st.y; cb = st.z; t = st.w;
        } else {
            t = edge_t[edge_slot];
        }
    }
    uint32_t shared_t = ~0u;  // :51
    bool left_early = false;  // the tile left the loop at a batch boundary (:66): nothing behind it is ever read
    // :62,66.  Batches end where the tile's WHOLE list reaches a multiple of 256 pairs; a launch that resumes a tile in
    // the middle of a batch (ROUND 2) first completes that batch.  The termination test belongs to those boundaries.
    for (int off = 0; off < num && shared_t > 255u;) {  // :66
        const int chunk = min(256 - (consumed & 255), num - off);  // :68
#ifdef GS_PROBE_TIMELINE
        const unsigned long long probe_p0 = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef GS_PROBE_TIMELINE
        const unsigned long long probe_p1 = __builtin_readcyclecounter();
        probe_tail += (uint32_t)(probe_p1 - probe_p0);
#endif
        // :72-75 staging (entries past the range are staged by nobody: nobody reads them)
        // Wave priority: a workgroup alternates between this section — a chain of gathers and ~600 VALU per staged splat,
        // then the list compaction — and the blend loop, which is pure issue; with eight workgroups per CU the SIMD's
        // arbiter would treat a wave about to request its slots like one in the middle of a blend list.  Staging and list
        // building run at priority 3, the blend loop at 0: requests leave earlier and the loop's waves fill the gaps.
        // Measured (profiles/r04_ab_call15_render_setprio_sweep.jsonl): launch 0.297 -> 0.288 ms at c3, 0.638 -> 0.610 ms
        // at c4; two frames in flight + 1.6 % / + 2.7 %.  Levels 1 and 2 for this section give half of that, dropping back
        // to 0 right after the gathers are requested costs 4 %, and the opposite assignment 1-3 %.
        __builtin_amdgcn_s_setprio(3);
        const bool have = (int)tid < chunk;
        if (have) {
            const uint32_t id = values[(size_t)bnd.x + (uint32_t)off + tid];
            if (DEG <= 0) {  // the projection pass of this frame wrote the records, colours included
                const float4 *r = culled + (size_t)id * 3;
                const float4 r0 = r[0], r1 = r[1], r2 = r[2];
                s_rec[tid * 3 + 0] = make_float4(r0.x, r0.y, (-0.5f * r1.x) * LOG2E, (-r1.y) * LOG2E);
                s_mask[tid] = (uint8_t)quadrant_mask(r0.x, r0.y, (0.5f * r1.x) * LOG2E, r1.y * LOG2E, (0.5f * r1.z) * LOG2E,
                                                     (float)(bx * TILE), (float)(by * TILE));
                s_rec[tid * 3 + 1].x = (-0.5f * r1.z) * LOG2E;
                s_rec[tid * 3 + 2] = r2;
            } else if constexpr (GEO) {
                const float4 *g = culled + (size_t)id * 2;
                const float4 *slot = sh_block + (size_t)(id - id_base) * SH_BLOCK_F4;
                const float4 g0 = g[0], g1 = g[1];
                const float4 pt = slot[SLOT_POS];
                s_rec[tid * 3 + 0] = g0;
                // (A, B, C) = (-hx, -hy, -hz): negation is exact, these are the eager branch's arguments bit for bit
                s_mask[tid] = (uint8_t)quadrant_mask(g0.x, g0.y, -g0.z, -g0.w, -g1.x, (float)(bx * TILE), (float)(by * TILE));
                float x, y, z, rgb[3];
                const float ms = rf.model_scale;   // (RasterizeData.pos = position * model_scale, splat_clip)
                sh_direction(pt.x * ms, pt.y * ms, pt.z * ms, rf.cam, x, y, z);
                sh_rgb<(DEG > 0 ? DEG : 1)>(slot, x, y, z, rgb);
                s_rec[tid * 3 + 1].x = g1.x;
                s_rec[tid * 3 + 2] = make_float4(rgb[0], rgb[1], rgb[2], g1.y);
            } else {
                // Lazy frame: no RasterizeData was written.  One gather of the splat's 256-byte scene slot (two whole
                // 128-byte lines: 48 SH coefficients + position, covariance, opacity) replaces the 48-byte record (1.25
                // lines on average) + the coefficient block (2 lines) of the round-2 build; the geometry half of the
                // record is recomputed here with the projection kernel's own expressions (project_math.h: ~200 VALU per
                // staged splat, noise next to the blend loop) — the values are the ones that kernel would have stored.
                const float4 *slot = sh_block + (size_t)(id - id_base) * SH_BLOCK_F4;
                const float4 pt = slot[SLOT_POS], A = slot[SLOT_COV_A], Bc = slot[SLOT_COV_B];
                const ClipPos cp = splat_clip(rf, pt);
                const Footprint ft = splat_footprint(rf, cp, pt.w, A, Bc);
                float ipx, ipy;
                splat_image_pos(rf, cp, ft.tf, ipx, ipy);
                float4 r0, r1;
                splat_raster_geometry(cp, ft, ipx, ipy, r0, r1);
                s_rec[tid * 3 + 0] = make_float4(r0.x, r0.y, (-0.5f * r1.x) * LOG2E, (-r1.y) * LOG2E);
                s_mask[tid] = (uint8_t)quadrant_mask(r0.x, r0.y, (0.5f * r1.x) * LOG2E, r1.y * LOG2E, (0.5f * r1.z) * LOG2E,
                                                     (float)(bx * TILE), (float)(by * TILE));
                // (Measured after the blend loop was rewritten, tools/render_timeline.py: a wave spends 29 % of its time in
                // this staging branch, 6 % at each of the barriers around it, 50 % in the blend loop.  Rearranging it
                // for latency — colour first so that both lines of the slot are requested together, every spilled loop
                // invariant rematerialised instead (no scratch reload left on the path), the next batch's ids requested
                // a batch ahead, one barrier less per batch — moved c3 by +3 %, c4 by -2 %: the phase is long because
                // its ~600 VALU instructions (eleven correctly rounded divisions among them) queue behind seven other
                // waves' on the SIMD, not because of its round trips.  The kernel is VALU-bound in every phase.)
                // the colour: channel after channel from the slot, 16 coefficient registers at a time (the whole kernel
                // stays at 64 VGPRs = 8 waves per SIMD).  Measured alternatives: 48 coefficients at once,
                // quad-cooperative loads through LDS, one colour channel per lane of a quad, a separate colour pass for
                // the splats staged in the previous frame — all slower.
                float x, y, z, rgb[3];
                sh_direction(r0.z, r0.w, r1.w, rf.cam, x, y, z);
                sh_rgb<(DEG > 0 ? DEG : 1)>(slot, x, y, z, rgb);
                s_rec[tid * 3 + 1].x = (-0.5f * r1.z) * LOG2E;
                s_rec[tid * 3 + 2] = make_float4(rgb[0], rgb[1], rgb[2], ft.opacity);
            }
        }
        if (tid == 0) s_sum = 0;  // :76
#ifdef GS_PROBE_TIMELINE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long probe_p2 = __builtin_readcyclecounter();
        probe_stage += (uint32_t)(probe_p2 - probe_p1);
#endif
        __syncthreads();
#ifdef GS_PROBE_TIMELINE
        const unsigned long long probe_p3 = __builtin_readcyclecounter();
        probe_bar2 += (uint32_t)(probe_p3 - probe_p2);
#endif

        // per-wave work list: the staged splats whose cutoff ellipse can reach this wave's quadrant (order kept)
        const int wave = (int)(tid >> 6);
        const uint32_t rec_lds = lds_address(s_rec);  // list entries are the records' LDS addresses
        int cnt = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int j = g * 64 + lane;
            const bool mine = j < chunk && ((s_mask[j] >> wave) & 1u);
            const unsigned long long m = __ballot(mine);
            if (mine) s_list[wave][cnt + (int)__popcll(m & ((1ull << lane) - 1ull))] = rec_lds + (uint32_t)(j * 48);
            cnt += (int)__popcll(m);
        }
        if (lane < 2) s_list[wave][cnt + lane] = rec_lds;  // the loop reads entries two steps ahead (and their records one)
        // (same wave wrote and reads s_list[wave]: LDS operations of one wave complete in order)

        // :79-91.  A lane whose pixel has reached t <= 1/255 has left the reference's loop (:79); a pixel below the
        // exp cutoff keeps colour and transmittance (alpha == 0, :86).  Both are the EXEC mask here, not arithmetic:
        // blend_list runs with exec = the alive lanes, narrows it to the lanes above the cutoff for the update and
        // retires pixels with v_cmpx — see there.
#ifdef GS_PROBE_TIMELINE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long probe_b0 = __builtin_readcyclecounter();
        probe_list += (uint32_t)(probe_b0 - probe_p3);
#endif
        __builtin_amdgcn_s_setprio(0);
        if (cnt > 0) blend_list<FAST_EXP>(lds_address(s_list[wave]), cnt, pxf, pyf, t, cr, cg, cb);
#ifdef GS_PROBE_TIMELINE
        const unsigned long long probe_b1 = __builtin_readcyclecounter();
        probe_blend += (uint32_t)(probe_b1 - probe_b0);
        probe_steps += (uint32_t)cnt;
#endif

        // :97 atomicAdd(shared_t, uint(t*255)) — integer sum, order-free
        uint32_t u = (uint32_t)(t * 255.0f);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) u += __shfl_xor(u, d, 64);
        if (lane == 0) atomicAdd(&s_sum, u);
        __syncthreads();
#ifdef GS_PROBE_TIMELINE
        probe_tail += (uint32_t)(__builtin_readcyclecounter() - probe_b1);
#endif
        off += chunk;
        consumed += chunk;
        // (a partial batch is the end of this launch's list: the loop ends on off == num; the sum is only looked at
        // where the reference looks at it, after a complete batch)
        shared_t = (consumed & 255) == 0 ? s_sum : ~0u;
        left_early = shared_t <= 255u;
    }

    // pairs of the tile staged so far = D_c of the reference's loop once the frame is complete (no atomics)
    if (tile_staged && tid == 0) tile_staged[tile_id] = (uint32_t)consumed;
    if constexpr (ROUND == 1) {
        if (!whole_frame) {
            if (tid == 0) {
                tile_done[tile_id] = left_early ? 1u : 0u;
                // (few tiles in a frame that is worth two rounds: round B's launches look at this count first)
                if (!left_early) atomicAdd(&plan->unfinished, 1u);
            }
            if (!left_early && tile_id == fp.gx * fp.gy - 1u) {  // T - 1 undecided: round 2 starts it over
                if (tid == 0) tile_staged[tile_id] = 0u;
                return;
            }
            if (!left_early) {  // unfinished: leave the state for round 2 (transmittance in the alpha channel)
                if (in_image)
                    image[(size_t)(pix_y - origin_y) * pitch_px + (pix_x - origin_x)] = make_float4(cr, cg, cb, t);
                else
                    edge_t[edge_slot] = t;
                return;
            }
        }
    }

    // :100-101
    const float a = (float)num * 5e-4f;
    const float h0 = 0.0f * (1.0f - a) + 1.0f * a;
    const float h1 = 0.0f * (1.0f - a) + 0.2f * a;
    const float h2 = 1.0f * (1.0f - a) + 0.2f * a;
    const float om = 1.0f - t;
    if (in_image) {
        image[(size_t)(pix_y - origin_y) * pitch_px + (pix_x - origin_x)] =
            make_float4(cr + (h0 * om) * rf.heatmap_factor, cg + (h1 * om) * rf.heatmap_factor,
                        cb + (h2 * om) * rf.heatmap_factor, 1.0f);
    }
#ifdef GS_PROBE_TIMELINE
    if (ROUND == 0 && (tid & 63u) == 0u) {  // pixels (0,0), (8,0), (0,8), (8,8) of the tile and their right neighbours
        const unsigned long long probe_t1 = __builtin_readcyclecounter();
        image[(size_t)(pix_y - origin_y) * pitch_px + (pix_x - origin_x)] =
            make_float4(__uint_as_float((uint32_t)probe_t0), __uint_as_float((uint32_t)(probe_t1 - probe_t0)),
                        __uint_as_float(__builtin_amdgcn_s_getreg(4 | (31 << 11)) & 0xFFFFu),
                        __uint_as_float(min(probe_blend, 0xFFFFFu) | (min(probe_steps, 4095u) << 20)));
        image[(size_t)(pix_y - origin_y) * pitch_px + (pix_x - origin_x) + 1] =
            make_float4(__uint_as_float(probe_stage), __uint_as_float(probe_bar2), __uint_as_float(probe_list), __uint_as_float(probe_tail));
    }
#endif
    // :105-110 picking.  subgroupElect() = first lane of each subgroup; the reference's sort pins the
    // subgroup width to 32, so "elected" = local pixel index (y*16+x) % 32 == 0, i.e. x == 0 and y even.
    if (ROUND == 0 && !BATCH && loc_x == 0u && (loc_y & 1u) == 0u && tile_id == fp.target_tile && t != 1.0f) {
        const uint32_t id = values[(size_t)bnd.x + (bnd.y - bnd.x) / 10u];
        if (DEG <= 0) {
            const float4 *r = culled + (size_t)id * 3;
            const float4 r0 = r[0], r1 = r[1];
            *pick = make_float4(r0.z, r0.w, r1.w, (float)num);
        } else {  // (lazy frame: RasterizeData.pos = position * model_scale, gsplat_projection.glsl:152,205)
            const ClipPos cp = splat_clip(fp, sh_block[(size_t)id * SH_BLOCK_F4 + SLOT_POS]);
            *pick = make_float4(cp.px, cp.py, cp.pz, (float)num);
        }
    }
}

}  // namespace

void launch_boundaries(const uint32_t *sorted_keys, const uint32_t *d_count, uint32_t num_tiles, uint2 *bounds,
                       bool fix_last_tile, bool sharded, const uint32_t *frame_last_tile_plus1, uint32_t *last_tile_keep,
                       const uint32_t *tie_values_in, uint32_t *tie_values_out, const uint32_t *tie_id_of,
                       uint32_t *long_count, uint32_t *long_list, uint32_t long_capacity, bool narrow_keys,
                       const TileMap &map, hipStream_t s) {
    if (last_tile_keep == frame_last_tile_plus1) last_tile_keep = nullptr;
    if (tie_values_in)
        hipLaunchKernelGGL(boundaries_ties_kernel, dim3(2048), dim3(256), 0, s, sorted_keys, d_count, num_tiles, bounds,
                           fix_last_tile ? 1 : 0, sharded ? 1 : 0, frame_last_tile_plus1, last_tile_keep, tie_values_in,
                           tie_values_out, tie_id_of, long_count, long_list, long_capacity);
    else if (narrow_keys)
        hipLaunchKernelGGL(boundaries_kernel<uint16_t>, dim3(2048), dim3(256), 0, s,
                           reinterpret_cast<const uint16_t *>(sorted_keys), d_count, num_tiles, bounds,
                           fix_last_tile ? 1 : 0, sharded ? 1 : 0, frame_last_tile_plus1, last_tile_keep, map);
    else
        hipLaunchKernelGGL(boundaries_kernel<uint32_t>, dim3(2048), dim3(256), 0, s, sorted_keys, d_count, num_tiles, bounds,
                           fix_last_tile ? 1 : 0, sharded ? 1 : 0, frame_last_tile_plus1, last_tile_keep, map);
}

void launch_tie_long_runs(uint32_t *keys_sorted, uint32_t *keys_scratch, uint32_t *values_in, uint32_t *values_out,
                          const uint32_t *d_count, const uint32_t *tie_id_of, uint32_t n_splats,
                          const uint32_t *long_count, const uint32_t *long_list, uint32_t long_capacity, hipStream_t s) {
    hipLaunchKernelGGL(tie_long_kernel, dim3(256), dim3(1024), 0, s, keys_sorted, keys_scratch, values_in, values_out,
                       d_count, tie_id_of, n_splats, long_count, long_list, long_capacity);
}

void launch_render(const float4 *culled, const float4 *sh_block, int lazy_degree, const uint32_t *sorted_values,
                   const uint2 *bounds, const FrameParams &fp, float4 *image, uint32_t image_pitch_px, uint32_t ox,
                   uint32_t oy, float4 *pick, uint32_t *tile_staged, const TileSchedule &sched, bool fast_exp,
                   hipStream_t s, int round, uint32_t *tile_done, FramePlan *plan, float *edge_t, bool geo) {
    if (fp.sx1 <= fp.sx0 || fp.sy1 <= fp.sy0) return;
    const uint32_t *tile_order = sched.order;
    const dim3 grid(tile_order ? sched.entries
                               : (fp.sx1 - fp.sx0) * (((fp.sy1 - fp.sy0) + 7u) / 8u) * 8u),  // rows rounded up to 8
        block(TILE, TILE);
#define GSPLAT_LAUNCH_RG(F, D, R, G)                                                                                       \
    hipLaunchKernelGGL((render_kernel<F, D, R, G, false>), grid, block, 0, s, culled, sh_block, sorted_values, bounds, fp, \
                       image, image_pitch_px, ox, oy, pick, tile_staged, tile_order, tile_done, plan, edge_t, NoBatch{})
#define GSPLAT_LAUNCH_R(F, D, R) \
    do { if (geo) GSPLAT_LAUNCH_RG(F, D, R, true); else GSPLAT_LAUNCH_RG(F, D, R, false); } while (0)
#define GSPLAT_LAUNCH_RD(F, R)                             \
    switch (d) {                                           \
        case 0: GSPLAT_LAUNCH_RG(F, 0, R, false); break;   \
        case 1: GSPLAT_LAUNCH_R(F, 1, R); break;           \
        case 2: GSPLAT_LAUNCH_R(F, 2, R); break;           \
        default: GSPLAT_LAUNCH_R(F, 3, R); break;          \
    }
    const int d = lazy_degree <= 0 ? 0 : (lazy_degree > 3 ? 3 : lazy_degree);
    if (fast_exp) {
        if (round == 1) { GSPLAT_LAUNCH_RD(true, 1) } else if (round == 2) { GSPLAT_LAUNCH_RD(true, 2) } else { GSPLAT_LAUNCH_RD(true, 0) }
    } else {
        if (round == 1) { GSPLAT_LAUNCH_RD(false, 1) } else if (round == 2) { GSPLAT_LAUNCH_RD(false, 2) } else { GSPLAT_LAUNCH_RD(false, 0) }
    }
#undef GSPLAT_LAUNCH_RD
#undef GSPLAT_LAUNCH_R
#undef GSPLAT_LAUNCH_RG
}

void launch_boundaries_batch(const uint32_t *sorted_keys, const uint32_t *d_count, uint2 *bounds, bool fix_last_tile,
                             bool sharded, const uint32_t *frame_last_tiles, const FrameBatch &batch, const TileMap &map,
                             hipStream_t s) {
    const FrameParams &f0 = batch.f[0];
    hipLaunchKernelGGL(boundaries_batch_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(sorted_keys),
                       d_count, bounds, fix_last_tile ? 1 : 0, sharded ? 1 : 0, frame_last_tiles, map, batch.rows, f0.sy0,
                       f0.gx * f0.gy);
}

void launch_render_batch(const float4 *records, const float4 *sh_block, int lazy_degree, const uint32_t *sorted_values,
                         const uint2 *bounds, const FrameParams &fpv, const FrameBatch &batch, float4 *image,
                         uint32_t image_pitch_px, uint32_t ox, uint32_t oy, uint32_t *tile_staged, const TileSchedule &sched,
                         bool fast_exp, bool geo, hipStream_t s) {
    if (fpv.sx1 <= fpv.sx0 || fpv.sy1 <= fpv.sy0) return;
    const uint32_t *tile_order = sched.order;
    const dim3 grid(tile_order ? sched.entries : (fpv.sx1 - fpv.sx0) * (((fpv.sy1 - fpv.sy0) + 7u) / 8u) * 8u), block(TILE, TILE);
#define GSPLAT_LAUNCH_B(F, D, G)                                                                                              \
    hipLaunchKernelGGL((render_kernel<F, D, 0, G, true>), grid, block, 0, s, records, sh_block, sorted_values, bounds, fpv,  \
                       image, image_pitch_px, ox, oy, static_cast<float4 *>(nullptr), tile_staged, tile_order,                \
                       static_cast<uint32_t *>(nullptr), static_cast<FramePlan *>(nullptr), static_cast<float *>(nullptr), batch)
#define GSPLAT_LAUNCH_BD(F)                                                                          \
    switch (d) {                                                                                     \
        case 0: GSPLAT_LAUNCH_B(F, 0, false); break;                                                 \
        case 1: if (geo) GSPLAT_LAUNCH_B(F, 1, true); else GSPLAT_LAUNCH_B(F, 1, false); break;     \
        case 2: if (geo) GSPLAT_LAUNCH_B(F, 2, true); else GSPLAT_LAUNCH_B(F, 2, false); break;     \
        default: if (geo) GSPLAT_LAUNCH_B(F, 3, true); else GSPLAT_LAUNCH_B(F, 3, false); break;    \
    }
    const int d = lazy_degree <= 0 ? 0 : (lazy_degree > 3 ? 3 : lazy_degree);
    if (fast_exp) { GSPLAT_LAUNCH_BD(true) } else { GSPLAT_LAUNCH_BD(false) }
#undef GSPLAT_LAUNCH_BD
#undef GSPLAT_LAUNCH_B
}

}  // namespace gsplat
