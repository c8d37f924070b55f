// Internal declarations shared by the HIP translation units of libgsplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsplat {

constexpr int TILE = 16;                 // gaussian_splatting_rasterizer.gd:4
constexpr int PROJ_BLOCK = 512;          // splats per projection workgroup (8 wave64): same kernel time as 256 on the same box, half the workgroup totals to scan (scan 31 -> 19 us at c3); 1024 loses occupancy
constexpr int SH_PLANES = 12;            // 48 SH floats as 12 float4 planes
constexpr int SH_BLOCK_F4 = 16;          // per-splat block for the compositor: 4 groups x {R, G, B, pad} float4

// Per-frame parameters handed to the kernels by value (the reference's uniform block + push constants).
struct FrameParams {
    float V[16];
    float P[16];
    float cam[3];
    float model_scale;
    float time;
    float Wf, Hf;          // float(dims)
    float Wm1, Hm1;        // float(dims - 1)
    uint32_t width, height;
    uint32_t gx, gy;       // tile grid
    uint32_t sx0, sx1, sy0, sy1; // stripe clamp in tiles
    float heatmap_factor;
    uint32_t target_tile;
    float view_norm2;      // upper bound of the squared spectral norm of the view matrix' 3x3 part (block culling)
    uint32_t cull_mode;    // 0 none, 1 workgroups outside a frustum plane, 2 also workgroups that cannot reach the stripe
};

// Scene in HBM, structure-of-arrays so that a wave's loads are 1 KiB contiguous per instruction and a
// culled splat costs 16 B instead of 240 B.
struct SceneSoA {
    float4 *pos_time;  // [N] x,y,z,load time
    float4 *cov_a;     // [N] xx,xy,xz,yy
    float4 *cov_b;     // [N] yz,zz,opacity,pad
    // the 48 SH floats (12 float4) of every splat, stored twice because two access patterns read them:
    float4 *sh_planes; // [12][N] plane-major: streamed by the projection pass when it evaluates the colours itself
    float4 *sh;        // [N][16] one 256-byte block per splat, gathered by the compositor when it evaluates the colour
                       // of the splats it stages: float4 4g+ch = coefficients 4g..4g+3 of channel ch (ch < 3), 4g+3
                       // unused: a channel's 16 coefficients are 4 loads, evaluated 16 registers at a time
};

struct SortBuffers {
    uint32_t *keys[2];
    uint32_t *values[2];
    uint32_t *part_hist;   // [max_partitions][RADIX]   (reduce-then-scan variant)
    uint32_t *digit_base;  // [RADIX] digit totals of the current pass (reduce-then-scan variant)
    // onesweep variant
    uint32_t small_count = 0;  // pair counts up to this use 1024-key partitions (sort.hip); 0 = never
    bool onesweep = false;
    uint32_t *global_hist = nullptr;  // [4][RADIX] digit totals of every pass
    uint32_t *status = nullptr;       // [4][max_partitions][RADIX] look-back words {flag:2 | count:30}
    uint32_t *tickets = nullptr;      // [4] partition ticket counters
    uint32_t *error_flag = nullptr;   // set if a look-back spin ran into its bound
};

// Optional per-launch timing: mark(k) records an event after a launch of kernel class k.
struct KernelTimer {
    static constexpr int MAX_MARKS = 64;
    hipEvent_t ev[MAX_MARKS + 1];
    int cls[MAX_MARKS];
    int count = 0;
    bool enabled = false;
    hipStream_t stream = nullptr;
    void begin(hipStream_t s) {
        count = 0;
        stream = s;
        if (enabled) (void)hipEventRecord(ev[0], s);
    }
    void mark(int k) {
        if (!enabled || count >= MAX_MARKS) return;
        cls[count] = k;
        (void)hipEventRecord(ev[++count], stream);
    }
};

// ---- launchers (each enqueues on `s`, no host sync) -------------------------------------------------
// sh_degree >= 0: the colours are evaluated here (plane-major SH); -1: left to the compositor
void launch_project(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                    uint32_t *local_off, uint32_t *counts, uint2 *rects, uint32_t *depths, uint4 *block_sums,
                    const float4 *block_bounds, uint32_t *block_skip, hipStream_t s);
// block_sums[b] = {pairs, visible splats, last tile + 1, 0} of workgroup b; block_bounds (nullable, 3 float4 per
// workgroup: {lo.xyz, max |cov|_F} {hi.xyz, max opacity factor} {latest load time,-,-,-}) + block_skip (u32 per
// workgroup, written by a small kernel launched first) enable fp.cull_mode
void launch_block_bounds(const SceneSoA &scene, uint32_t n, float4 *block_bounds, hipStream_t s);
// fused projection + emission (default); chunk_status/chunk_info have project_num_chunks(n) entries
void launch_project_emit(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                         uint32_t *counts, unsigned long long *chunk_status, uint32_t *ticket, uint2 *chunk_info,
                         uint64_t capacity, uint32_t *keys, uint32_t *values, uint64_t *total_out, uint32_t *d_sorted,
                         uint32_t *overflow, uint32_t *visible_out, uint32_t *last_tile_out, uint32_t *error_flag,
                         hipStream_t s);
uint32_t project_num_chunks(uint32_t n);
// split variant: scan of the workgroup totals (also finalises D, min(D,capacity), overflow, V, last tile), then emit
void launch_scan_blocks(const uint4 *block_sums, uint32_t num_blocks, uint64_t *block_base, uint64_t capacity,
                        uint64_t *total_out, uint32_t *d_sorted, uint32_t *overflow, uint32_t *visible_out,
                        uint32_t *last_tile_out, uint2 *bounds, uint32_t bounds_entries, uint32_t *big_count,
                        const uint32_t *tile_staged, uint32_t num_tiles, uint32_t *host_hint, hipStream_t s);
// host_hint (nullable, host-mapped): {visible splats of this frame, pairs the compositor staged last frame, frames}  // also clears bounds and the big-rectangle list counter
void launch_emit(uint32_t n, const FrameParams &fp, const uint32_t *local_off, const uint32_t *counts,
                 const uint2 *rects, const uint32_t *depths, const uint4 *block_sums, const uint64_t *block_base,
                 uint64_t capacity, uint32_t *keys, uint32_t *values, uint32_t *big_count, uint32_t *big_list,
                 hipStream_t s);  // big_list: emit_big_list_entries(capacity) ids of splats covering > 512 tiles
uint32_t emit_big_list_entries(uint64_t capacity);

// Stable LSD radix sort of (key,value) pairs on the low `sig_bits` bits.  The element count is read
// from device memory (*d_count), never from the host.  Returns the index (0/1) of the buffer pair that
// holds the sorted result.
int launch_sort_pairs(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int sig_bits, hipStream_t s,
                      KernelTimer *kt = nullptr, int first_bit = 0);
// tile-major sort, second half (tilesort.hip): stable sort of every tile's segment [segs[t].x, segs[t].y) on the low
// 16 key bits, in place in (keys_a, vals_a); (keys_b, vals_b) is scratch for segments longer than 4096 pairs
// big_count (one device word, zero at launch) / big_list (num_tiles words): tiles with more than 4096 pairs
int launch_tile_depth_sort(uint32_t *keys_a, uint32_t *vals_a, uint32_t *keys_b, uint32_t *vals_b, const uint2 *segs,
                           uint32_t num_tiles, const uint32_t *d_count, uint32_t *big_count, uint32_t *big_list,
                           hipStream_t s);
int sort_num_passes(int sig_bits);
uint32_t sort_max_partitions(uint64_t capacity);
uint32_t sort_small_count_default();

// tie_* non-null (re-laid-out scene): the same pass also restores the order of equal keys to ascending splat id
// (values hold storage slots; tie_id_of[slot] = splat id) and writes the result to tie_values_out
// segs (nullable): the tiles' true segments [first, end) without the quirks of bounds
void launch_boundaries(const uint32_t *sorted_keys, const uint32_t *d_count, uint32_t num_tiles, uint2 *bounds,
                       uint2 *segs, bool fix_last_tile, bool sharded, const uint32_t *frame_last_tile_plus1,
                       const uint32_t *tie_values_in, uint32_t *tie_values_out, const uint32_t *tie_id_of,
                       hipStream_t s);
// sh_degree >= 0: the compositor evaluates the SH colour of the splats it stages from scene_sh (sh_eval.h);
// -1: RasterizeData already holds the colours
void launch_render(const float4 *culled, const float4 *scene_sh, int sh_degree, const uint32_t *sorted_values,
                   const uint2 *bounds, const FrameParams &fp, float4 *image, uint32_t image_pitch_px, uint32_t origin_x,
                   uint32_t origin_y, float4 *pick, uint32_t *tile_staged, bool fast_exp, hipStream_t s);
// tile_staged[tile] = pairs staged (D_c); pixel (x,y) -> image[(y-origin_y)*pitch + (x-origin_x)]
void launch_fill_colors(float4 *culled, const float4 *scene_sh, int sh_degree, const uint32_t *counts, uint32_t n,
                        const FrameParams &fp, hipStream_t s);  // parity tap: colour of every splat that emitted pairs

// scene ingest
void launch_upload_records(const SceneSoA &scene, uint32_t n_total, uint32_t first, uint32_t count,
                           const float *d_records, uint32_t *sh_degree_max, const uint32_t *slot_of, hipStream_t s);
void launch_upload_ply_rows(const SceneSoA &scene, uint32_t n_total, uint32_t first, uint32_t count,
                            const float *d_rows, float load_time, uint32_t *sh_degree_max, const uint32_t *slot_of,
                            hipStream_t s);
void launch_gather_records(const SceneSoA &scene, uint32_t n_total, float *d_records, const uint32_t *slot_of,
                           hipStream_t s);
// scene re-layout and the taps that undo it
void launch_permute_float4(const float4 *src, float4 *dst, const uint32_t *id_of, uint32_t n, uint32_t rec,
                           hipStream_t s);  // records of `rec` float4s
void launch_gather_u32(const uint32_t *src, uint32_t *dst, const uint32_t *index, uint32_t n, hipStream_t s);
void launch_gather_raster(const float4 *culled, float4 *dst, const uint32_t *slot_of, uint32_t n, hipStream_t s);

}  // namespace gsplat
