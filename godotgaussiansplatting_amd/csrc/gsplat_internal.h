// Internal declarations shared by the HIP translation units of libgsplat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsplat {

constexpr int TILE = 16;                 // gaussian_splatting_rasterizer.gd:4
constexpr int PROJ_BLOCK = 512;          // splats per projection workgroup (8 wave64): same kernel time as 256 on the same box, half the workgroup totals to scan; 1024 loses occupancy
constexpr int SH_BLOCK_F4 = 16;          // per-splat 256-byte slot: 3 channels x 4 float4 of SH coefficients (192 B), then
                                         // copies of the splat's pos_time, cov_a, cov_b (SLOT_POS ..) and 16 B of padding
constexpr int SLOT_POS = 12, SLOT_COV_A = 13, SLOT_COV_B = 14;
// the compositor's heaviest-first tile schedule: ORDER_LPT's single list is built by one workgroup, for stripes of up
// to ORDER_MAX_TILES tiles (beyond that: static row order); ORDER_XCD's eight lists by one workgroup each, for lists of up
// to ORDER_MAX_SLOTS slots (the XCD-local schedule pads partial blocks) — every grid of up to 65 536 tiles
constexpr uint32_t ORDER_MAX_TILES = 16384;
constexpr uint32_t ORDER_MAX_SLOTS = 18432;

// Compositor schedules (raster.hip: which tile workgroup b takes)
enum : uint32_t {
    ORDER_ROWS = 0,  // static: tile row r of the stripe -> XCD r % 8, no table
    ORDER_LPT = 1,   // one list, heaviest tile first (previous frame's staged counts)
    ORDER_XCD = 2,   // eight interleaved lists, one per XCD (workgroup b runs on XCD b % 8): heaviest first inside a list
};
// ORDER_XCD: the stripe is cut into blocks of bw x bh tiles (8 x 2: neighbours gather many of the same splat records,
// horizontally and vertically) and block B of the row-major block grid belongs to XCD B % 8 — so every XCD's L2 sees
// compact pieces from all over the stripe (balanced cost) and all the tiles that share a record with a given tile
// inside its block.  nbx is kept off the multiples of 8 by one virtual, always empty block column: otherwise every
// block column would belong to one XCD.  The table has 8 * per_xcd slots, slot 8 * j + x = j-th heaviest tile of XCD x;
// slots of partial / virtual blocks hold 0xFFFFFFFF and sort last.
struct OrderLayout {
    uint32_t bw, bh, nbx, nby, nblocks, per_xcd, entries;
};
__host__ __device__ inline OrderLayout order_layout(uint32_t sw, uint32_t sh) {
    OrderLayout l;
    l.bw = sw < 8u ? (sw ? sw : 1u) : 8u;
    l.bh = sh < 2u ? 1u : 2u;
    l.nbx = (sw + l.bw - 1u) / l.bw;
    if ((l.nbx & 7u) == 0u) l.nbx += 1u;
    l.nby = (sh + l.bh - 1u) / l.bh;
    l.nblocks = l.nbx * l.nby;
    l.per_xcd = ((l.nblocks + 7u) / 8u) * (l.bw * l.bh);
    l.entries = 8u * l.per_xcd;
    return l;
}
inline size_t order_capacity(uint32_t gx, uint32_t gy) { return (size_t)(gx + 15u) * (gy + 1u) + 112u; }
struct TileSchedule {
    uint32_t *order = nullptr;  // nullptr: ORDER_ROWS
    uint32_t entries = 0;       // workgroups of the compositor launch = table slots
    uint32_t mode = ORDER_ROWS;
};
#ifndef GSPLAT_SPLAT_PART
#define GSPLAT_SPLAT_PART 2048
#endif
constexpr int SPLAT_PART0 = GSPLAT_SPLAT_PART;  // slots per partition of the splat-sort passes (a multiple of PROJ_BLOCK)

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
    // camera constants of project_covariance (gsplat_projection.glsl:127-133), evaluated once per frame on the host in
    // IEEE binary32 exactly as the shader would per splat — uniform values belong in scalar registers, not in every
    // lane's VGPRs: tan_fov = 1 / P00, 1 / P11; focal0 = (dims * 0.5) * (P00, P11); lim = tan_fov * 1.3
    float tan_x, tan_y, focal0_x, focal0_y, lim_x, lim_y;
    float view_norm2;      // upper bound of the squared spectral norm of the view matrix' 3x3 part (block culling)
    uint32_t cull_mode;    // 0 none, 1 workgroups outside a frustum plane, 2 also workgroups that cannot reach the stripe
};

// B frames of one context through ONE launch sequence (gsplat_render_batch; DESIGN.md §6 "batched frames").  A stripe rank
// of an 8-GPU frame runs fourteen launches of which ten take the same 5 - 15 us whatever their input, and its compositor
// is bound by the serial chain of its heaviest tile (340 tiles on 256 CUs): both are per LAUNCH, not per frame.  A batch
// renders B consecutive frames — B cameras, one scene — as ONE frame of a virtual image that stacks the B stripes
// vertically: virtual splat b * n_pad + slot is slot's splat seen by camera b, its tile rectangle lands in rows
// [b * rows, (b + 1) * rows) of a grid of gx x (B * rows) tiles.  Everything between the projection and the compositor
// (splat sort, emission, pair sort — the stripe-local 16-bit tile ids now run to B * stripe_tiles) sees one larger frame and
// is unchanged; only the kernels that touch a CAMERA or a REAL pixel / tile id know about the batch: block culling and
// projection (camera b), the scan (a "last tile" per frame), the tile ranges (quirk Q5/Q6 belongs to each real frame) and
// the compositor (colours: camera position b; pixels: image b).
constexpr int MAX_BATCH = 4;
struct FrameBatch {
    FrameParams f[MAX_BATCH];   // the REAL frames: camera, clock, real grid and stripe
    uint32_t count;             // B
    uint32_t n_pad;             // slots per frame in the virtual index space (N rounded up to a multiple of PROJ_BLOCK)
    uint32_t blocks;            // projection workgroups per frame = n_pad / PROJ_BLOCK
    uint32_t rows;              // tile rows of the stripe = rows of the virtual grid per frame
    uint32_t image_stride_px;   // pixels between the images of consecutive frames (the compositor's target)
};
struct NoBatch {};              // the kernel argument of the plain instantiations
// the frame a workgroup works for: its own slice of the batch, or the launch's one frame
template <bool BATCH, typename B>
__host__ __device__ __forceinline__ const FrameParams &frame_of(const FrameParams &fp, const B &batch, uint32_t b) {
    if constexpr (BATCH) return batch.f[b];
    else return fp;
}

// Scene in HBM, structure-of-arrays so that a wave's loads are 1 KiB contiguous per instruction and a
// culled splat costs 16 B instead of 240 B.  272 B per splat (the reference's AoS record: 240 B).
struct SceneSoA {
    float4 *pos_time;  // [N] x,y,z,load time
    float4 *cov_a;     // [N] xx,xy,xz,yy
    float4 *cov_b;     // [N] yz,zz,opacity,pad
    float4 *sh_dc;     // [N] band 0: coefficient 0 of R, G, B (+ pad) — all a degree-0 scene ever reads (streamed)
    float4 *sh_block;  // nullptr while the scene has only seen band-0 colours (then nothing reads it: 256 N bytes saved);
                       // [N][16] one 256-byte, 256-byte-aligned slot per splat = exactly two 128-byte lines: all 48
                       // coefficients, channel-major (float4 4*ch + g = coefficients 4g .. 4g+3 of channel ch), then a
                       // copy of pos_time / cov_a / cov_b.  Everything the compositor of a lazy frame needs for a splat
                       // it stages — it recomputes the screen-space record (project_math.h) and evaluates the colour
                       // from this one gather; the streaming projection kernel keeps reading the SoA planes above
};

// 16-bit pair keys are STRIPE-LOCAL tile ids: lid = (ty - sy0) * sw + (tx - sx0), monotone in the tile id inside the
// stripe (both row-major), so the sorted order, the tile ranges and every tap are what global ids would give — but a
// stripe of an 8-GPU frame (or a small frame) then has few enough key bits for ONE pair pass (sort.hip, "wide" pass).
// A full-frame context: sx0 = sy0 = 0, sw = gx, lid = tile id.  32-bit keys (tile << 16 | depth16) stay global.
struct TileMap {
    uint32_t gx, sx0, sy0, sw;
    __host__ __device__ uint32_t local_of(uint32_t tx, uint32_t ty) const { return (ty - sy0) * sw + (tx - sx0); }
    __host__ __device__ uint32_t global_of(uint32_t lid) const {
        const uint32_t r = lid / sw;
        return (sy0 + r) * gx + sx0 + (lid - r * sw);
    }
};
inline TileMap tile_map_of(const FrameParams &fp) {
    const uint32_t sw = fp.sx1 > fp.sx0 ? fp.sx1 - fp.sx0 : 1u;
    return TileMap{fp.gx, fp.sx0, fp.sy0, sw};
}

// Hand-off of the projection pass, indexed by storage slot (splat id in an un-finalized scene).
struct SplatKeys {
    uint32_t *key;   // depth16 | (tile id of the rectangle's origin) << 16; defined where dims != 0
    uint32_t *dims;  // w | h << 16 of the tile rectangle (already clamped to the stripe); 0 = the splat emits nothing
};
// Compact list of the splats that emit pairs, in (depth16, slot) order after the two splat-level radix passes.
struct SplatList {
    uint32_t *key, *id, *dims;
};

// Device-side plan of a two-round frame (projection.hip: frame_plan_kernel)
struct FramePlan {
    uint32_t v_a;     // round A composites the first v_a entries of the depth-sorted splat list
    uint32_t single;  // 1: round A is the whole frame (one round after all: D exceeds the key budget)
    uint32_t unfinished;  // tiles round A's compositor left unfinished (counted by it; zeroed by frame_plan_kernel): 0 =
                          // round B has nothing to do and its launches leave at once (tile_sat_kernel skips the table)
    uint32_t pad;
};
// two-round frames are used up to this many tiles (the unfinished-tile table is u16 and built in LDS)
constexpr uint32_t ROUNDS_MAX_TILES = 32768;
// ... whose worst case is a 1 x 32768 grid: 2 x 32769 entries
constexpr size_t ROUNDS_MAX_SAT_BYTES = (size_t)2 * (ROUNDS_MAX_TILES + 1) * sizeof(uint16_t);

struct SortBuffers {
    uint32_t *keys[2];
    uint32_t *values[2];
    uint32_t *part_hist;   // [RADIX][max_partitions], digit-major
    uint32_t *digit_base;  // [RADIX] digit totals of the current pass
    uint32_t *wide_hist = nullptr;  // one-pass form of the pair level (sort.hip "wide" pass): count matrix
                                    // [WIDE_MAX_PARTS][bins] + bins digit totals; allocated on first use (api.hip)
    uint32_t wide_bins_allocated = 0;
    uint32_t small_count = 0;  // element counts up to this use 1024-key partitions (sort.hip); 0 = never
    bool rank_atomic = false;  // downsweeps rank with returning LDS atomics (set once sort_rank_selftest() has passed)
    // splat-level passes (depth16 of the visible splats)
    SplatList list[2];
    uint32_t *splat_hist;  // [RADIX][ceil(N/512)] pass 0 (written by the projection kernel), reused by pass 1
    uint32_t *v_count;     // number of splats that emit pairs this frame (device)
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
// sh_degree >= 0: the colours are evaluated here for every visible splat (0: from the band-0 plane, streamed; 1..3:
// from the splat's coefficient block); -1: left to the compositor (RasterizeData.color = 0), no record written;
// -2: colours left to the compositor, and `culled` receives the 32-byte STAGED geometry of every visible splat (2 float4
// per slot: project_math.h staged_geometry) instead of RasterizeData
// block_sums[b] = {pairs, visible splats, last tile + 1, skipped} of workgroup b; block_bounds (nullable, 3 float4 per
// workgroup: {lo.xyz, max |cov|_F} {hi.xyz, max opacity factor} {latest load time,-,-,-}) + block_skip (u32 per
// workgroup, written by a small kernel launched first) enable fp.cull_mode.  splat_hist: this workgroup's 256-bin
// histogram of (depth16 & 255) over its visible splats = pass 0 of the splat sort, digit-major.
void launch_project(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                    const SplatKeys &keys, uint4 *block_sums, uint32_t *splat_hist, const float4 *block_bounds,
                    uint32_t *block_skip, const uint32_t *tile_staged, uint32_t num_tiles, uint32_t *dc_parts,
                    const TileSchedule &sched, hipStream_t s, uint32_t *live = nullptr, uint32_t blocks_per_part = 4);
// live (nullable; used when the frame culls blocks): live_list_words(blocks) words — the compact lists of the blocks that
// were NOT skipped and of the partitions of splat-sort pass 0 (blocks_per_part = sort_splat_part_blocks(n) blocks each) that
// hold one; the projection workgroups and pass 0 (launch_sort_splats) are dealt from these lists, XCD by XCD
inline size_t live_list_words(size_t blocks) { return 8 + 2 * blocks; }
uint32_t sort_splat_part_blocks(uint32_t n);   // projection workgroups per partition of the splat sort's pass 0 for n slots
// (tile_staged .. sched: extra workgroups at the front of the launch order the stripe's tiles for the compositor by what it
// staged for them in the previous frame — one per XCD list — and leave that frame's D_c in dc_parts[0..8), which
// launch_scan_blocks adds up for the host; tile_staged == nullptr: no extra workgroups)
// batch form (FrameBatch above): fpv = the VIRTUAL frame (gx x count * rows tiles, stripe = all of it); keys / block_sums /
// splat_hist / block_skip / records are indexed by virtual slot resp. virtual workgroup (count * batch.blocks of them)
void launch_project_batch(const SceneSoA &scene, uint32_t n, const FrameBatch &batch, const FrameParams &fpv, int sh_degree,
                          float4 *records, const SplatKeys &keys, uint4 *block_sums, uint32_t *splat_hist,
                          const float4 *block_bounds, uint32_t *block_skip, const uint32_t *tile_staged, uint32_t num_tiles,
                          uint32_t *dc_parts, const TileSchedule &sched, hipStream_t s, uint32_t *live = nullptr,
                          uint32_t blocks_per_part = 4);
void launch_block_bounds(const SceneSoA &scene, uint32_t n, float4 *block_bounds, hipStream_t s);
void launch_pow02_bits(uint32_t first_bits, uint64_t count, float *out, hipStream_t s);  // parity tap of pow(x, 0.2)
// parity tap: the RasterizeData record of EVERY visible splat of the frame `fp` (a lazy frame writes none)
void launch_fill_records(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                         hipStream_t s);
// pairs per 512-splat block of the sorted splat list (the block-local offsets are recomputed by the emit kernel)
void launch_emit_sums(const SplatList &list, const uint32_t *v_count, uint32_t n, uint32_t *emit_sums, hipStream_t s);
// scan of the block totals: block_base (64-bit), D / min(D, capacity) / overflow / visible / frame's last tile;
// also clears tile_bounds, the big-rectangle list counter and the long-run list counter (long_count).
// host_hint (nullable, host-mapped): {visible splats of this frame, pairs the compositor staged last frame, frames posted,
// big rectangles met since the last posting (big_seen folds the emissions that scan without posting)};
// pairs_hint (nullable, host-mapped): receives min(D, capacity) of this call; last_tile_copy (nullable, device): a second
// home for the frame's "last tile + 1" (gsplat_render_begin's word)
void launch_scan_blocks(const uint32_t *emit_sums, const uint4 *proj_sums, uint32_t num_blocks, uint64_t *block_base,
                        uint64_t capacity, uint64_t *total_out, uint32_t *d_sorted, uint32_t *overflow,
                        uint32_t *visible_out, uint32_t *last_tile_out, uint2 *bounds, uint32_t bounds_entries,
                        uint32_t *big_count, uint32_t *host_hint, const uint32_t *dc_parts, uint32_t *pairs_hint,
                        uint32_t *last_tile_copy, uint32_t *long_count, uint32_t *big_seen, hipStream_t s,
                        uint32_t frame_blocks = 0);
// frame_blocks != 0 (batched frames): workgroups [b * frame_blocks, (b + 1) * frame_blocks) belong to frame b, and
// last_tile_out / last_tile_copy receive one word PER FRAME (MAX_BATCH words each)
// two-round frames (projection.hip)
void launch_frame_plan(const uint4 *proj_sums, uint32_t num_blocks, uint64_t capacity, uint32_t frac16,
                       uint64_t *total_out, FramePlan *plan, uint32_t *d_hint, hipStream_t s);  // d_hint: host-mapped, nullable
// (re-laid-out scenes) round A ends where the depth code of the sorted list changes
void launch_plan_align(const uint32_t *list_key, const uint32_t *v_count, FramePlan *plan, hipStream_t s);
size_t tile_sat_entries(uint32_t gx, uint32_t gy);
int launch_tile_sat(const uint32_t *tile_done, const FramePlan *plan, const FrameParams &fp, uint16_t *sat, hipStream_t s);
void launch_round_filter(const SplatList &list, const uint32_t *v_count, uint32_t n, const FramePlan *plan,
                         const uint16_t *sat, const uint32_t *tile_done, const FrameParams &fp, uint32_t *key_out,
                         uint32_t *dims_out, uint32_t *emit_sums, hipStream_t s);
// host_hint (nullable, host-mapped): {visible splats of this frame, pairs the compositor staged last frame, frames,
// frame counter}
void launch_emit(const SplatList &list, const uint32_t *v_count, uint32_t n, const FrameParams &fp,
                 const uint32_t *emit_sums, const uint64_t *block_base, uint64_t capacity, uint32_t *keys,
                 uint32_t *values, uint32_t *big_count, uint32_t *big_list, bool narrow_keys, hipStream_t s,
                 uint32_t split = 1, bool list_bigs = true, uint32_t big_hint = 0);
// big_hint: how many rectangles of more than 512 tiles the context's recent emissions met (sizes the second launch's grid)
// big_list: 2 words per entry; split: workgroups per 512-entry block of the list; list_bigs: rectangles of more than 512
// tiles are listed and written by a second launch in which the whole grid shares each of them — false: no second launch,
// the wave that owns such a rectangle writes it itself (they are still COUNTED in *big_count: the host posts that count
// back and asks for the second launch in the frames that follow a frame that met any)
uint32_t emit_big_list_entries(uint64_t capacity);

// Splat-level half of the sort: the visible splats ordered by (depth16, slot) — two stable 8-bit passes over
// 12-byte elements; pass 0 reads the projection hand-off (its histograms come from the projection kernel) and compacts.
// Result in sb.list[0]; *sb.v_count = number of elements.
// block_skip (nullable): per projection workgroup, 1 = culled this frame — it wrote neither rectangle sizes nor its
// histogram column, and both are taken as zero here.
void launch_sort_splats(SortBuffers &sb, const SplatKeys &keys, uint32_t n, const uint32_t *block_skip, hipStream_t s,
                        KernelTimer *kt = nullptr, const uint32_t *live = nullptr);
// live (nullable, with block_skip): the frame's live lists (launch_project) — pass 0 walks the list of live partitions
// Pair-level half: stable LSD radix passes over the key bits [first_bit, sig_bits) of (key,value) pairs.  The
// element count is read from device memory (*d_count), never from the host.  Returns the index (0/1) of the buffer
// pair that holds the result.
// narrow_keys: sb.keys[] hold 16-bit tile ids instead of the reference's 32-bit (tile << 16 | depth16) keys — the
// depth half orders nothing at the pair level; bit b of the wide key is bit b - 16 of the narrow one.
int launch_sort_pairs(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, int sig_bits, hipStream_t s,
                      KernelTimer *kt = nullptr, int first_bit = 0, bool narrow_keys = false);
// keys_out[i] = keys16[i] << 16 | depth16 of splat values[i] (taps of a narrow-key frame; splat_keys = SplatKeys::key)
// The pair level in ONE pass over 16-bit stripe-local tile ids below `bins` (1024 or 4096; sort_wide_bins(tiles), 0 = the
// stripe has too many — or too few — tiles for it): input keys[0] / values[0], result in half 1 (returned).  Needs
// sb.wide_hist of sort_wide_hist_words(bins) words.  Any pair count is sorted correctly; the form pays off for small ones.
constexpr uint32_t WIDE_MAX_PARTS = 1024;
uint32_t sort_wide_bins(uint32_t tiles);
size_t sort_wide_hist_words(uint32_t bins);
int launch_sort_pairs_wide(SortBuffers &sb, const uint32_t *d_count, uint64_t capacity, uint32_t bins, hipStream_t s,
                           KernelTimer *kt = nullptr);
void launch_widen_keys(const uint16_t *keys16, const uint32_t *values, const uint32_t *splat_keys,
                       const uint32_t *d_count, uint32_t *keys_out, const TileMap &map, hipStream_t s);
int sort_num_passes(int sig_bits);
uint32_t sort_max_partitions(uint64_t capacity);
uint32_t sort_small_count_default();
bool sort_rank_selftest();  // true: same-address LDS atomics of a wave come back in lane order on this device

// Tile ranges (gsplat_boundaries.glsl).  tie_* non-null (re-laid-out scene): the same pass also restores the order of
// equal keys to ascending splat id (values hold storage slots; tie_id_of[slot] = splat id) and writes the result to
// tie_values_out; runs of more than 64 equal keys are listed (long_count / long_list) and sorted by
// launch_tie_long_runs.  keys_scratch / values_in are clobbered inside such runs (keys_sorted is restored).
// last_tile_keep (nullable): receives a copy of *frame_last_tile_plus1 — the word a replay of the frame asks with
void launch_boundaries(const uint32_t *sorted_keys, const uint32_t *d_count, uint32_t num_tiles, uint2 *bounds,
                       bool fix_last_tile, bool sharded, const uint32_t *frame_last_tile_plus1, uint32_t *last_tile_keep,
                       const uint32_t *tie_values_in, uint32_t *tie_values_out, const uint32_t *tie_id_of,
                       uint32_t *long_count, uint32_t *long_list, uint32_t long_capacity, bool narrow_keys,
                       const TileMap &map, hipStream_t s);
// batched frames (16-bit stripe-local keys of the virtual frame): quirk Q5/Q6 applies to every REAL frame's highest
// populated tile — frame_last_tiles: batch.count words
void launch_boundaries_batch(const uint32_t *sorted_keys, const uint32_t *d_count, uint2 *bounds, bool fix_last_tile,
                             bool sharded, const uint32_t *frame_last_tiles, const FrameBatch &batch, const TileMap &map,
                             hipStream_t s);
void launch_tie_long_runs(uint32_t *keys_sorted, uint32_t *keys_scratch, uint32_t *values_in, uint32_t *values_out,
                          const uint32_t *d_count, const uint32_t *tie_id_of, uint32_t n_splats,
                          const uint32_t *long_count, const uint32_t *long_list, uint32_t long_capacity, hipStream_t s);
// lazy_degree >= 1: the compositor evaluates the SH colour (bands 0..lazy_degree) of the splats it stages from
// sh_block (sh_eval.h); 0: RasterizeData already holds the colours
void launch_render(const float4 *culled, const float4 *sh_block, int lazy_degree, const uint32_t *sorted_values,
                   const uint2 *bounds, const FrameParams &fp, float4 *image, uint32_t image_pitch_px, uint32_t origin_x,
                   uint32_t origin_y, float4 *pick, uint32_t *tile_staged, const TileSchedule &sched, bool fast_exp,
                   hipStream_t s, int round = 0, uint32_t *tile_done = nullptr, FramePlan *plan = nullptr,
                   float *edge_t = nullptr, bool geo = false);
// geo (lazy_degree >= 1 only): `culled` is the frame's STAGED-GEOMETRY buffer — 2 float4 per storage slot, written by
// launch_project(sh_degree = -2) — and the compositor gathers those instead of recomputing the projection of what it stages
// batched frames: fpv = the virtual frame (schedule, tile ranges and staged counts are indexed by virtual tile), the pixels of
// frame b go to image + b * batch.image_stride_px; one round, no pick
void launch_render_batch(const float4 *records, const float4 *sh_block, int lazy_degree, const uint32_t *sorted_values,
                         const uint2 *bounds, const FrameParams &fpv, const FrameBatch &batch, float4 *image,
                         uint32_t image_pitch_px, uint32_t origin_x, uint32_t origin_y, uint32_t *tile_staged,
                         const TileSchedule &sched, bool fast_exp, bool geo, hipStream_t s);
// round 1 / 2: the two launches of a two-round frame (tile_done: round 1 marks the tiles it finished; plan: device;
// edge_t: (gx + gy) x 256 floats, the transmittance of the out-of-image lanes of unfinished edge tiles between the rounds)
// tile_staged[tile] = pairs staged (D_c); pixel (x,y) -> image[(y-origin_y)*pitch + (x-origin_x)]
// parity tap; block_skip (nullable): the frame's per-block cull marks — a skipped block wrote no rectangle sizes
void launch_tile_counts(const uint32_t *dims, uint32_t *counts, uint32_t n, const uint32_t *block_skip, hipStream_t s);

// scene ingest
void launch_upload_records(const SceneSoA &scene, uint32_t n_total, uint32_t first, uint32_t count,
                           const float *d_records, uint32_t *sh_degree_max, const uint32_t *slot_of, hipStream_t s);
void launch_upload_ply_rows(const SceneSoA &scene, uint32_t n_total, uint32_t first, uint32_t count,
                            const float *d_rows, float load_time, uint32_t *sh_degree_max, const uint32_t *slot_of,
                            hipStream_t s);
void launch_gather_records(const SceneSoA &scene, uint32_t n_total, float *d_records, const uint32_t *slot_of,
                           hipStream_t s);
void launch_build_slots(const SceneSoA &scene, uint32_t n, hipStream_t s);  // sh_block of a band-0 scene, from its planes
// scene re-layout and the taps that undo it
// 30-bit Morton code of every position inside the box of the finite positions (box6: 6 words of scratch) + ids 0..n-1
void launch_morton_keys(const float4 *pos, uint32_t n, uint32_t *box6, uint32_t *codes, uint32_t *ids, hipStream_t s);
void launch_invert_permutation(const uint32_t *id_of_slot, uint32_t n, uint32_t *slot_of_id, hipStream_t s);
void launch_permute_float4(const float4 *src, float4 *dst, const uint32_t *id_of, uint32_t n, uint32_t rec,
                           hipStream_t s);  // records of `rec` float4s
void launch_gather_u32(const uint32_t *src, uint32_t *dst, const uint32_t *index, uint32_t n, hipStream_t s);
void launch_gather_raster(const float4 *culled, float4 *dst, const uint32_t *slot_of, uint32_t n, hipStream_t s);

}  // namespace gsplat

// what group.hip needs to know about a context (api.hip)
struct gsplat_ctx;
struct gsplat_frame;
namespace gsplat {
struct CtxView {
    int device;
    hipStream_t stream;
    uint32_t width, height, gx, gy;
    float4 *image;          // the default target (context-owned or imported)
    bool timing;
    bool stripe_cull;       // frames begun with gsplat_render_begin skip blocks that cannot reach the stripe: the frame's
                            // last tile then has to be exchanged (GSPLAT_FLAG_BLOCK_CULL on a finalized scene)
    bool ties_storage;      // GSPLAT_FLAG_TIES_STORAGE_ORDER: the members of a group must agree on it
};
CtxView ctx_view(gsplat_ctx *c);
void ctx_record_gather(gsplat_ctx *c, hipEvent_t start, hipEvent_t stop);  // -> gsplat_stats.ms_gather (events owned by the group)
void ctx_set_last_image(gsplat_ctx *c, float4 *image);                      // the image tap follows group frames
bool ctx_join_group(gsplat_ctx *c, const void *group);                     // nullptr: leave; false: already in another group
// gsplat_render_begin with the caller's word on stripe culling: false = blocks are skipped against the frustum only, so the
// context's own "last tile" is the frame's (a group whose ranks agreed not to exchange it)
int ctx_render_begin(gsplat_ctx *c, const gsplat_frame *frame, uint32_t *last_tile_out_device, bool stripe_cull);
// batched frames (api.hip): gsplat_render_batch_begin with the caller's word on stripe culling; frames a context's buffers hold
int ctx_batch_begin(gsplat_ctx *c, const gsplat_frame *frames, uint32_t count, uint32_t *last_tiles_out_device, bool stripe_cull);
uint32_t ctx_batch_capacity(const gsplat_ctx *c);
int set_last_error(const char *text, int status);                          // thread-local detail for gsplat_last_error
}  // namespace gsplat
