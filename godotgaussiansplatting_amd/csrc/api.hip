// C ABI of libgsplat_hip.so (include/gsplat.h): scene store, contexts, device memory, frame orchestration.
// Host-side counterpart of util/gaussian_splatting_rasterizer.gd (init_gpu / rasterize /
// get_splat_position / texture_size setter / update_camera_matrices) with HIP streams and device-side
// counters instead of Vulkan descriptor sets, indirect dispatches and per-dispatch barriers.
//
// Two kinds of state:
//   SceneStore  the splat buffer (gaussian_splatting_rasterizer.gd:83), its optional Morton re-layout and the
//               ingest machinery (upload stream + pinned staging ring) — shared by every context created on it;
//   gsplat_ctx  one output size / stripe / stream and the intermediate buffers of a frame (RasterizeData, the sort
//               buffers, tile ranges, image): frames in flight and the stripes of one GPU are several contexts on one
//               scene (gsplat_create_view), one upload and one copy of the scene in HBM.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include <unistd.h>

#include "../../include/gsplat.h"
#include "gsplat_internal.h"
#include "rounds_controller.h"

// lazy frames: the projection kernel writes the staged geometry of every visible splat and the compositor gathers it
// (1), or the compositor recomputes it from the scene for the pairs it stages (0).  Decided by measurement.
#ifndef GSPLAT_GEO_DEFAULT
#define GSPLAT_GEO_DEFAULT 0
#endif
// frames that cull blocks: projection workgroups and splat-sort pass 0 are dealt from compact lists of the live blocks /
// partitions (1) or walk the index range and leave where a block was skipped (0)
#ifndef GSPLAT_LIVE_LISTS_DEFAULT
#define GSPLAT_LIVE_LISTS_DEFAULT 0
#endif

using namespace gsplat;

namespace gsplat {
// group.hip: the three colour channels of a w x h rectangle of a row-major RGBA32F image (pitch in pixels), packed — the
// stripes of a multi-GPU frame travel like that, and so does a frame read back with GSPLAT_FLAG_READBACK_RGB
void launch_pack_rgb(const float4 *image, uint32_t pitch, uint32_t w, uint32_t h, float *packed_rgb, hipStream_t s);
}  // namespace gsplat

namespace {

thread_local char g_last_error[512] = "";

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    snprintf(g_last_error, sizeof g_last_error, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return e == hipErrorOutOfMemory ? GSPLAT_ERR_OUT_OF_MEMORY : GSPLAT_ERR_HIP;
}

#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) return hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

// device counters of a context, one 64-byte block
struct Counters {
    uint64_t total_emitted;  // D before the clamp
    uint32_t d_sorted;       // min(D, capacity): the pair count every later pass reads
    uint32_t overflow;
    uint32_t visible;        // splats that wrote RasterizeData (sum over the projection workgroups)
    uint32_t frame_last_tile_plus1;  // highest tile touched by any splat's unclamped rectangle, +1
    uint32_t big_count;      // splats listed for emit_big_kernel this frame
    uint32_t long_count;     // runs of more than 64 equal keys listed for tie_long_kernel this frame
    uint32_t v_count;        // elements of the sorted splat list (= splats that emit pairs in this context's stripe)
    uint32_t hint_frames;    // frames whose {V, D_c} the scan kernel has posted to the host (big_count[3]; never cleared)
    FramePlan plan;          // two-round frames: size of round A, or "one round after all" (frame_plan_kernel)
    uint64_t round_total[2]; // pairs emitted by round A / B of a two-round frame
    uint32_t round_overflow; // (round B's scan writes its always-false overflow flag here, not over the frame's)
    uint32_t replay_last_tile_plus1;  // the frame's last tile + 1 as the last frame's boundaries pass saw it
    uint32_t dc_parts[8];    // the previous frame's D_c, one part per schedule workgroup of the projection launch
    uint32_t big_seen;       // most big rectangles an emission met since the count was last posted to the host
    uint32_t pad[3];
    uint32_t batch_last_tile[MAX_BATCH];  // batched frames: every frame's highest tile touched + 1 (scan_blocks_kernel)
};

// the one-pass pair sort is taken (policy auto) while the previous frame emitted at most this many pairs: above, its
// scattered stores and the count matrix cost more than the second pass of the split form saves (sort.hip)
constexpr uint32_t WIDE_AUTO_PAIRS = 4u << 20;

constexpr int STAGING_SLOTS = 4;
constexpr size_t STAGING_BYTES = 8u << 20;  // per slot: 33 k .ply rows / 34 k records per piece

struct StagingSlot {
    std::mutex mutex;            // one uploader at a time fills and submits a slot
    void *host = nullptr;        // pinned
    void *dev = nullptr;
    hipEvent_t done = nullptr;   // the kernel that consumed the slot's last piece
    bool used = false;
};

}  // namespace

struct gsplat_ctx;

namespace {

struct SceneStore {
    int device = 0;
    uint32_t n = 0;
    uint32_t num_proj_blocks = 0;
    SceneSoA soa{};
    // gsplat_finalize_scene: storage slot <-> splat id, per-workgroup bounds for block culling
    bool finalized = false;
    uint32_t *id_of_slot = nullptr, *slot_of_id = nullptr;
    float4 *block_bounds = nullptr;         // written on upload_stream only (finalize, uploads to a finalized scene): every
                                            // frame is ordered behind upload_done, so no view ever reads them half-made
    // ingest
    hipStream_t upload_stream = nullptr;
    StagingSlot ring[STAGING_SLOTS];
    std::atomic<uint32_t> next_slot{0};
    std::mutex mutex;               // upload_done event, views list, finalize
    hipEvent_t upload_done = nullptr;  // recorded on upload_stream at the end of every upload call
    bool any_upload = false;
    bool bounds_dirty = false;         // uploads reached a finalized scene since block_bounds was last taken (under mutex)
    uint32_t *deg_host = nullptr, *deg_dev = nullptr;  // host-mapped word: atomicMax target of the upload kernels
    std::atomic<int> sh_degree_seen{0};
    uint64_t bytes = 0;
    std::vector<void *> allocations;
    std::vector<gsplat_ctx *> views;

    ~SceneStore() {
        (void)hipSetDevice(device);
        if (upload_stream) {
            (void)hipStreamSynchronize(upload_stream);
            (void)hipStreamDestroy(upload_stream);
        }
        for (void *p : allocations) (void)hipFree(p);
        for (auto &sl : ring) {
            if (sl.host) (void)hipHostFree(sl.host);
            if (sl.dev) (void)hipFree(sl.dev);
            if (sl.done) (void)hipEventDestroy(sl.done);
        }
        if (upload_done) (void)hipEventDestroy(upload_done);
        if (deg_host) (void)hipHostFree(deg_host);
    }
};

void raise_degree(SceneStore *sc, int deg) {
    int cur = sc->sh_degree_seen.load();
    while (deg > cur && !sc->sh_degree_seen.compare_exchange_weak(cur, deg)) {}
}

}  // namespace

struct gsplat_ctx {
    gsplat_config cfg{};
    int device = 0;
    std::shared_ptr<SceneStore> scene;
    hipStream_t stream = nullptr;
    bool own_stream = false;

    uint32_t n = 0;
    // batched frames (gsplat_create_batch_view; gsplat_internal.h FrameBatch): the intermediate buffers hold `batch` frames —
    // per-splat arrays batch * n_pad virtual slots, the pair buffers batch * the frame's key budget, per-tile arrays and the
    // image batch * the frame's.  batch == 1: a plain context
    uint32_t batch = 1, n_pad = 0;
    uint32_t last_batch = 0;           // frames of the last gsplat_render_batch (0: the last frame was a plain one)
    FrameBatch front_batch{};          // the batch gsplat_render_batch_begin started ...
    FrameParams front_fpv{};           // ... and its virtual frame
    uint64_t capacity = 0;
    uint32_t width = 0, height = 0, gx = 0, gy = 0;
    uint32_t sx0 = 0, sx1 = 0, sy0 = 0, sy1 = 0;  // stripe in tiles

    float4 *culled = nullptr;          // RasterizeData[N]: allocated by the first frame that writes it (an eager frame) or
                                       // the first tap that asks for it — a context that only renders lazy frames (every
                                       // scene with SH bands above 0, by default) never holds these 48 N bytes
    float4 *geo = nullptr;             // staged geometry, 2 float4 per slot (project_math.h staged_geometry): written by the
                                       // projection kernel of a geometry-eager lazy frame, gathered by its compositor;
                                       // allocated by the first such frame
    bool keys_wide = false;            // sort.keys[] hold `capacity` 32-bit keys; otherwise `capacity` 16-bit tile ids
    SplatKeys keys{};
    uint4 *block_sums = nullptr;       // per projection workgroup: pairs, visible, last tile + 1, skipped
    uint32_t *emit_sums = nullptr;
    uint64_t *block_base = nullptr;
    uint32_t *big_list = nullptr;      // list positions of splats covering > 512 tiles, written by emit_big_kernel
    uint32_t *long_list = nullptr;     // first elements of runs of > 64 equal keys (finalized scenes)
    uint32_t long_capacity = 0;
    uint32_t *block_skip = nullptr;    // per frame: 1 = the projection workgroup cannot emit anything
    uint32_t *live_lists = nullptr;    // per frame with block culling: the workgroups / pass-0 partitions NOT skipped, compact
                                       // (projection.hip live_lists_kernel): what projection and pass 0 are dealt from
    bool use_live_lists = GSPLAT_LIVE_LISTS_DEFAULT != 0;  // GSPLAT_LIVE_LISTS=on|off (A/B, tests; same outputs)
    SortBuffers sort{};
    uint32_t *emit_keys = nullptr, *emit_values = nullptr;  // GSPLAT_FLAG_KEEP_EMITTED
    uint2 *bounds = nullptr;
    uint32_t *tile_staged = nullptr;
    uint32_t *tile_order = nullptr;    // compositor schedule: the stripe's tiles, heaviest (previous frame) first
    uint32_t order_mode = ORDER_XCD;   // GSPLAT_TILE_ORDER=rows|lpt|xcd: the compositor's schedule (A/B; same image)
    float4 *image = nullptr;
    float4 *pick = nullptr;
    Counters *counters = nullptr;
    // hand-off of finished frames (gsplat.h: gsplat_bind_external_image, gsplat_render_async)
    hipExternalMemory_t ext_mem = nullptr;   // imported allocation of the host's graphics API ...
    float4 *ext_image = nullptr;             // ... and the image inside it: the default target while bound
    float4 *last_image = nullptr;            // context-owned target of the last frame (the image tap reads it)
    struct AsyncRing {
        float4 *dev[2] = {nullptr, nullptr};          // the ring's own two device images
        float *dev_rgb[2] = {nullptr, nullptr};       // GSPLAT_FLAG_READBACK_RGB: their colour channels, packed (what is copied)
        float *host[3] = {nullptr, nullptr, nullptr}; // pinned: RGBA32F, or RGB32F with that flag
        hipEvent_t rendered[2] = {nullptr, nullptr}, copy_start[3] = {nullptr, nullptr, nullptr},
                   copy_done[3] = {nullptr, nullptr, nullptr};
        hipStream_t stream = nullptr;
        uint64_t count = 0;                           // frames submitted with gsplat_render_async
        uint64_t last_waited = 0;
        bool ready = false;
    } async;
    hipEvent_t gather_start = nullptr, gather_stop = nullptr;  // owned by the context's group (gsplat_group_render)
    const void *group = nullptr;       // the gsplat_group this context is a member of: it caches the context's size and
                                       // pointer, so gsplat_resize / gsplat_destroy refuse until the group is destroyed
    uint64_t bytes_allocated = 0;

    // where the SH colours are evaluated this frame: by the compositor for the splats it stages (lazy) or by the
    // projection pass for every visible splat (eager).  Chosen per frame from what the previous frames did.
    int color_policy = 0;              // 0 auto, 1 always lazy, 2 always eager (GSPLAT_COLOR)
    bool front_lazy = false, last_lazy = false;
    // lazy frames: does the projection kernel hand the compositor the staged geometry of every visible splat (32 B per
    // splat written, 32 B per staged pair gathered), or does the compositor recompute it from the scene for what it stages?
    int geo_policy = GSPLAT_GEO_DEFAULT;  // 1: geometry-eager lazy frames, 0: the compositor recomputes (GSPLAT_GEO=on|off: A/B, tests)
    bool front_geo = false, last_geo = false;
    // the pair-level buffers of the frame hold 16-bit tile ids instead of 32-bit keys (sort.hip): whenever the scene is
    // in upload order (the tie repair of a re-laid-out scene compares whole keys)
    bool front_narrow = false, last_narrow = false;
    // Two-round frames (projection.hip): round A composites the front rounds_frac16 / 65536 of the depth-sorted splats,
    // round B what their unfinished tiles still need.  The sorted-pair taps, tile_bounds and the pick of such a frame
    // are produced on demand by replaying it in one round (replay_full).
    bool front_rounds = false, last_rounds = false;
    bool taps_stale = false;           // the sort buffers hold round B's arrays: taps and pick replay the frame first
    int rounds_policy = 0;             // 0 auto (controller: choose_rounds), 1 never, 2 pinned fraction (GSPLAT_ROUNDS)
    uint32_t rounds_frac16 = 16384;    // size of round A as a fraction of the visible splats, x 65536
    // controller: timed trials of settings (one round / two rounds with a fraction), rounds_controller.h
    struct RoundsSlot { hipEvent_t start = nullptr, end = nullptr; bool pending = false, counts = false; uint32_t trial = 0; };
    RoundsSlot rounds_ring[4];
    int rounds_slot = -1;              // ring slot timing the frame now between render_front and render_back
    int rounds_next_slot = 0;
    RoundsController rounds_ctl;
    gsplat_frame last_frame{};         // for the replay
    bool front_stripe_cull = false, last_stripe_cull = false;
    bool front_skip_marks = false, last_skip_marks = false;  // the frame ran with block culling: block_skip holds its marks
    bool front_list_bigs = false;      // this frame's emissions list rectangles of more than 512 tiles for emit_big_kernel
    uint32_t front_big_hint = 0;       // ... and how many the recent emissions met (sizes that launch)
    uint32_t *tile_done = nullptr;     // round A: 1 = the tile left its loop at a batch boundary (finished)
    uint16_t *tile_sat = nullptr;      // summed-area table of the unfinished tiles, (gy + 1) x (gx + 1)
    float *edge_t = nullptr;           // transmittance of the out-of-image lanes of unfinished edge tiles, between the rounds
    bool wide_keys_only = false;       // GSPLAT_KEYS=wide (A/B, tests)
    int pair_sort_policy = 0;          // 0 auto, 1 always the split passes, 2 the one-pass form wherever the stripe allows
                                       // (GSPLAT_PAIR_SORT=split|wide: A/B and tests; same sorted pairs)
    uint32_t front_wide_bins = 0, last_wide_bins = 0;  // bins of the frame's one-pass pair sort (0: split passes)
    bool ties_storage = false;         // GSPLAT_FLAG_TIES_STORAGE_ORDER: equal keys stay in storage order in a re-laid-out
                                       // scene — no tie repair, hence no need for whole keys at the pair level
    uint32_t *hint_host = nullptr;     // host-mapped: {visible splats, pairs staged by the previous frame, frames}
    uint32_t *hint_dev = nullptr;      // the same words as the device sees them

    int sorted_index = 0;  // which ping-pong half holds the sorted pairs (keys) of the last frame
    int values_index = 0;  // ... and the sorted values (differs from sorted_index after the tie fix-up)
    SceneSoA front_soa{}, last_soa{};  // the scene's arrays as the begun / last frame saw them (wait_for_uploads)
    int bigs_unknown = 3;  // frames for which this context cannot know yet whether its emissions meet rectangles of more than
                           // 512 tiles (new context, new stripe / size / layout): they run WITH the listed second launch
    FrameParams front_fp;  // parameters of the frame gsplat_render_begin started
    FrameParams last_fp;   // parameters of the last finished frame (parity taps)
    bool front_done = false;
    int front_sig_bits = 0, front_sh_degree = 0;
    int last_sig_bits = 32, last_sh_degree = 0;
    bool rendered = false;
    hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timing_valid = false;
    KernelTimer kt;
    bool kt_events_created = false;

#ifdef GSPLAT_TEST_HOOKS
    // diagnosis builds only (experiments/README.md, "asymmetric CU masks"): the compositor of this context's frames on a
    // stream of its own that may only use GSPLAT_PROBE_RENDER_CUS of the 256 compute units, so that the byte-bound kernels
    // of ANOTHER context's frame find compute units the issue-bound compositor does not hold
    hipStream_t probe_render_stream = nullptr;
    hipEvent_t probe_render_ready = nullptr, probe_render_done = nullptr;
#endif

    std::vector<void *> allocations;
};

namespace {

template <typename T>
int raw_alloc(std::vector<void *> &list, uint64_t &bytes_total, T **out, size_t count, bool zero, hipStream_t s) {
    const size_t bytes = (count ? count : 1) * sizeof(T);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    list.push_back(p);
    bytes_total += bytes;
    if (zero) HIP_TRY(hipMemsetAsync(p, 0, bytes, s));
    *out = static_cast<T *>(p);
    return GSPLAT_OK;
}

template <typename T>
int dev_alloc(gsplat_ctx *c, T **out, size_t count, bool zero) {
    return raw_alloc(c->allocations, c->bytes_allocated, out, count, zero, c->stream);
}

void dev_release(gsplat_ctx *c, void *p, size_t bytes) {
    if (!p) return;
    for (size_t i = 0; i < c->allocations.size(); ++i)
        if (c->allocations[i] == p) {
            c->allocations.erase(c->allocations.begin() + i);
            break;
        }
    c->bytes_allocated -= bytes;
    (void)hipFree(p);
}

int apply_stripe(gsplat_ctx *c, uint32_t axis, uint32_t b, uint32_t e) {
    if (axis == GSPLAT_STRIPE_NONE) {
        c->sx0 = 0; c->sx1 = c->gx; c->sy0 = 0; c->sy1 = c->gy;
    } else if (axis == GSPLAT_STRIPE_COLUMNS) {
        if (b > e || e > c->gx) return GSPLAT_ERR_OUT_OF_RANGE;
        c->sx0 = b; c->sx1 = e; c->sy0 = 0; c->sy1 = c->gy;
    } else if (axis == GSPLAT_STRIPE_ROWS) {
        if (b > e || e > c->gy) return GSPLAT_ERR_OUT_OF_RANGE;
        c->sx0 = 0; c->sx1 = c->gx; c->sy0 = b; c->sy1 = e;
    } else {
        return GSPLAT_ERR_INVALID_ARGUMENT;
    }
    c->cfg.stripe_axis = axis; c->cfg.stripe_begin = b; c->cfg.stripe_end = e;
    return GSPLAT_OK;
}

struct SizeBuffers {
    uint2 *bounds = nullptr;
    uint32_t *tile_staged = nullptr;
    uint32_t *tile_order = nullptr;
    uint32_t *tile_done = nullptr;
    uint16_t *tile_sat = nullptr;
    float *edge_t = nullptr;
    float4 *image = nullptr;
};

// the heaviest-first tile schedule of this frame's stripe, or nullptr = static row order (too many tiles, or switched off)
TileSchedule scheduled_tiles(const gsplat_ctx *c, const FrameParams &fp) {
    TileSchedule t;
    const uint32_t sw = fp.sx1 - fp.sx0, sh = fp.sy1 - fp.sy0;
    const uint64_t stripe_tiles = (uint64_t)sw * sh;
    if (stripe_tiles == 0) return t;
    if (c->order_mode == ORDER_LPT && stripe_tiles <= ORDER_MAX_TILES) {
        t.order = c->tile_order; t.entries = (uint32_t)stripe_tiles; t.mode = ORDER_LPT;
    } else if (c->order_mode == ORDER_XCD) {
        // eight workgroups order one XCD list each (schedule_tiles): what has to fit their LDS is ONE list, so the
        // XCD-local schedule covers every grid the library accepts (4K: 8 lists of 4080 slots; round 3 stopped at 16 384
        // tiles in all, and 4K frames ran the static row order that cost 11 % at 1080p)
        const OrderLayout lay = order_layout(sw, sh);
        if (lay.per_xcd <= ORDER_MAX_SLOTS && lay.entries <= order_capacity(c->gx, c->gy * c->batch)) {
            t.order = c->tile_order; t.entries = lay.entries; t.mode = ORDER_XCD;
        }
    }
    return t;
}

size_t bounds_entries(uint32_t gx, uint32_t gy) { return ((size_t)gx * gy + 1) & ~(size_t)1; }

int alloc_size_dependent(gsplat_ctx *c, uint32_t width, uint32_t height, uint32_t gx, uint32_t gy, SizeBuffers *out) {
    int rc;
    gy *= c->batch;      // (a batch context: the virtual grid stacks `batch` stripes of up to gy rows; `batch` images)
    height *= c->batch;
    if ((rc = dev_alloc(c, &out->bounds, bounds_entries(gx, gy), true))) return rc;
    if ((rc = dev_alloc(c, &out->tile_staged, (size_t)gx * gy, true))) return rc;
    if ((rc = dev_alloc(c, &out->tile_order, order_capacity(gx, gy), true))) return rc;
    if ((rc = dev_alloc(c, &out->tile_done, (size_t)gx * gy, true))) return rc;
    if ((rc = dev_alloc(c, &out->tile_sat, tile_sat_entries(gx, gy), true))) return rc;
    if ((rc = dev_alloc(c, &out->edge_t, (size_t)(gx + gy) * 256, true))) return rc;
    if ((rc = dev_alloc(c, &out->image, (size_t)width * height, true))) return rc;
    return GSPLAT_OK;
}

void release_size_dependent(gsplat_ctx *c, const SizeBuffers &b, uint32_t width, uint32_t height, uint32_t gx,
                            uint32_t gy) {
    gy *= c->batch;
    height *= c->batch;
    dev_release(c, b.bounds, bounds_entries(gx, gy) * sizeof(uint2));
    dev_release(c, b.tile_staged, (size_t)gx * gy * sizeof(uint32_t));
    dev_release(c, b.tile_order, order_capacity(gx, gy) * sizeof(uint32_t));
    dev_release(c, b.tile_done, (size_t)gx * gy * sizeof(uint32_t));
    dev_release(c, b.tile_sat, tile_sat_entries(gx, gy) * sizeof(uint16_t));
    dev_release(c, b.edge_t, (size_t)(gx + gy) * 256 * sizeof(float));
    dev_release(c, b.image, (size_t)width * height * sizeof(float4));
}

bool is_device_pointer(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

void fill_frame_params(const gsplat_ctx *c, const gsplat_frame *f, FrameParams *fp) {
    memcpy(fp->V, f->view, sizeof fp->V);
    memcpy(fp->P, f->proj, sizeof fp->P);
    fp->cam[0] = f->cam_pos[0]; fp->cam[1] = f->cam_pos[1]; fp->cam[2] = f->cam_pos[2];
    fp->model_scale = f->model_scale;
    fp->time = f->time;
    fp->Wf = (float)c->width; fp->Hf = (float)c->height;
    fp->Wm1 = (float)((int)c->width - 1); fp->Hm1 = (float)((int)c->height - 1);
    fp->width = c->width; fp->height = c->height;
    fp->gx = c->gx; fp->gy = c->gy;
    fp->sx0 = c->sx0; fp->sx1 = c->sx1; fp->sy0 = c->sy0; fp->sy1 = c->sy1;
    fp->heatmap_factor = f->heatmap_factor;
    fp->target_tile = f->target_tile;
    {   // gsplat_projection.glsl:127-133 per-frame constants, binary32 like the shader's (this file is built with
        // -ffp-contract=off; volatile keeps the host compiler from folding the chain in a wider type)
        volatile float tix = f->proj[0], tiy = f->proj[5];
        volatile float hx = fp->Wf * 0.5f, hy = fp->Hf * 0.5f;
        volatile float tfx = 1.0f / tix, tfy = 1.0f / tiy;
        fp->focal0_x = hx * tix; fp->focal0_y = hy * tiy;
        fp->tan_x = tfx; fp->tan_y = tfy;
        fp->lim_x = tfx * 1.3f; fp->lim_y = tfy * 1.3f;
    }
    // |W|_2^2 of the view matrix' 3x3 part, bounded by Gershgorin on W^t W (1 for a rigid camera): block culling
    double g[3][3], bound = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            g[i][j] = 0.0;
            for (int r = 0; r < 3; ++r) g[i][j] += (double)f->view[i * 4 + r] * (double)f->view[j * 4 + r];
        }
    for (int i = 0; i < 3; ++i) bound = std::max(bound, std::fabs(g[i][0]) + std::fabs(g[i][1]) + std::fabs(g[i][2]));
    fp->view_norm2 = (float)(bound * 1.00001);
    fp->cull_mode = 0;
}

int sig_bits_for(uint32_t tiles) {
    int bits = 0;
    while ((1u << bits) < tiles) ++bits;
    return 16 + bits;
}

// ---- scene -------------------------------------------------------------------------------------------------------
int scene_create(int device, uint32_t n, std::shared_ptr<SceneStore> *out) {
    std::shared_ptr<SceneStore> sc(new (std::nothrow) SceneStore());
    if (!sc) return GSPLAT_ERR_OUT_OF_MEMORY;
    sc->device = device;
    sc->n = n;
    sc->num_proj_blocks = (uint32_t)(((size_t)n + PROJ_BLOCK - 1) / PROJ_BLOCK);
    HIP_TRY(hipStreamCreateWithFlags(&sc->upload_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&sc->upload_done, hipEventDisableTiming));
    int rc;
    hipStream_t s = sc->upload_stream;
    // gaussian_splatting_rasterizer.gd:83, re-laid out as SoA: 272 B per splat
    if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->soa.pos_time, (size_t)n, true, s))) return rc;
    if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->soa.cov_a, (size_t)n, true, s))) return rc;
    if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->soa.cov_b, (size_t)n, true, s))) return rc;
    if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->soa.sh_dc, (size_t)n, true, s))) return rc;
    // (soa.sh_block — 256 B per splat — only once a colour band above 0 shows up: ensure_slots)
    for (auto &sl : sc->ring) {
        HIP_TRY(hipHostMalloc(&sl.host, STAGING_BYTES, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&sl.dev, STAGING_BYTES));
        HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        sc->bytes += STAGING_BYTES;
    }
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sc->deg_host), 64, hipHostMallocMapped));
    memset(sc->deg_host, 0, 64);
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&sc->deg_dev), sc->deg_host, 0));
    HIP_TRY(hipStreamSynchronize(s));
    *out = sc;
    return GSPLAT_OK;
}

// highest SH band with a non-zero coefficient (NaN counts as non-zero, -0.0 as zero), like the upload kernels
int host_degree_records(const float *rec, uint32_t count) {
    int deg = 0;
    for (uint32_t i = 0; i < count && deg < 3; ++i) {
        const float *sh = rec + (size_t)i * GSPLAT_RECORD_FLOATS + 12;
        for (int k = 47; k >= 3; --k)
            if (sh[k] != 0.0f) {
                deg = std::max(deg, k < 12 ? 1 : (k < 27 ? 2 : 3));
                break;
            }
    }
    return deg;
}

int host_degree_rows(const float *rows, uint32_t count) {
    int deg = 0;
    for (uint32_t i = 0; i < count && deg < 3; ++i) {
        const float *rest = rows + (size_t)i * GSPLAT_PLY_ROW_FLOATS + 9;  // f_rest: channel-major, 15 per channel
        for (int ch = 0; ch < 3; ++ch)
            for (int k = 14; k >= 0; --k)
                if (rest[15 * ch + k] != 0.0f) {
                    deg = std::max(deg, k < 3 ? 1 : (k < 8 ? 2 : 3));
                    break;
                }
    }
    return deg;
}

// The 256-byte gather slots (SceneSoA::sh_block) exist only in scenes that carry SH bands above 0: a band-0 scene (c2,
// c5: every frame eager, 16 B of colour streamed per splat) never reads them, and they are 3/4 of a scene's bytes.  The
// first upload that brings a higher band — or a frame that is told to evaluate one — allocates them and fills the slots
// of everything uploaded so far from the planes (higher coefficients zero: what the upload kernels would have written).
// Caller holds sc->mutex; the fill runs on the upload stream, ahead of the upload kernel that needs the slots and of
// upload_done, which every frame waits for.
int ensure_slots(SceneStore *sc) {
    if (sc->soa.sh_block != nullptr) return GSPLAT_OK;
    float4 *slots = nullptr;
    const int rc = raw_alloc(sc->allocations, sc->bytes, &slots, (size_t)sc->n * SH_BLOCK_F4, false, sc->upload_stream);
    if (rc != GSPLAT_OK) return rc;
    SceneSoA with = sc->soa;
    with.sh_block = slots;
    launch_build_slots(with, sc->n, sc->upload_stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sc->upload_done, sc->upload_stream));
    sc->any_upload = true;
    sc->soa.sh_block = slots;
    return GSPLAT_OK;
}

int upload_common(gsplat_ctx *c, uint32_t first, uint32_t count, const float *src, int floats_per_item, bool ply_rows,
                  float load_time) {
    if (!c || (!src && count)) return GSPLAT_ERR_INVALID_ARGUMENT;
    SceneStore *sc = c->scene.get();
    if ((uint64_t)first + count > sc->n) return GSPLAT_ERR_OUT_OF_RANGE;
    if (!count) return GSPLAT_OK;
    HIP_TRY(hipSetDevice(sc->device));
    hipStream_t us = sc->upload_stream;
    const uint32_t *slot_of = sc->finalized ? sc->slot_of_id : nullptr;
    if (is_device_pointer(src)) {
        // the caller's device buffer is read in place; the band count comes back through a host-mapped word (so the slots
        // have to exist before the kernel runs: what it will find is not known here)
        {
            std::lock_guard<std::mutex> lock(sc->mutex);
            const int rc = ensure_slots(sc);
            if (rc != GSPLAT_OK) return rc;
            if (ply_rows) launch_upload_ply_rows(sc->soa, sc->n, first, count, src, load_time, sc->deg_dev, slot_of, us);
            else launch_upload_records(sc->soa, sc->n, first, count, src, sc->deg_dev, slot_of, us);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipStreamSynchronize(us));  // this stream only: rendering goes on
        raise_degree(sc, (int)*reinterpret_cast<volatile uint32_t *>(sc->deg_host));
    } else {
        // host memory: pieces go through the pinned staging ring — memcpy, async H2D copy, kernel; the only wait is
        // for the ring slot's own previous piece.  Nothing here allocates, frees or synchronises the device.
        const size_t item_bytes = (size_t)floats_per_item * sizeof(float);
        const uint32_t piece_max = (uint32_t)(STAGING_BYTES / item_bytes);
        int deg = 0;
        for (uint32_t done = 0; done < count; done += piece_max) {
            const uint32_t m = std::min(count - done, piece_max);
            const float *piece = src + (size_t)done * floats_per_item;
            StagingSlot &sl = sc->ring[sc->next_slot.fetch_add(1u) % STAGING_SLOTS];
            std::lock_guard<std::mutex> lock(sl.mutex);
            if (sl.used) HIP_TRY(hipEventSynchronize(sl.done));
            memcpy(sl.host, piece, (size_t)m * item_bytes);
            const int piece_deg = ply_rows ? host_degree_rows(piece, m) : host_degree_records(piece, m);
            deg = std::max(deg, piece_deg);
            {   // (the scene's layout — slots or none — is read and the kernel enqueued under the scene's lock: an uploader
                // that brings the first higher band builds the slots of everything enqueued before it, and only of that)
                std::lock_guard<std::mutex> lock(sc->mutex);
                if (piece_deg > 0) {
                    const int rc = ensure_slots(sc);
                    if (rc != GSPLAT_OK) return rc;
                }
                HIP_TRY(hipMemcpyAsync(sl.dev, sl.host, (size_t)m * item_bytes, hipMemcpyHostToDevice, us));
                const float *d_src = static_cast<const float *>(sl.dev);
                if (ply_rows) launch_upload_ply_rows(sc->soa, sc->n, first + done, m, d_src, load_time, sc->deg_dev, slot_of, us);
                else launch_upload_records(sc->soa, sc->n, first + done, m, d_src, sc->deg_dev, slot_of, us);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(sl.done, us));
            }
            sl.used = true;
        }
        raise_degree(sc, deg);
    }
    {   // frames submitted after this call returns are ordered behind it (stream-side wait, no host wait)
        std::lock_guard<std::mutex> lock(sc->mutex);
        // the stored scene changed: the block bounds of a finalized scene are stale.  They are retaken ONCE per batch of
        // uploads, by the next frame (wait_for_uploads: on the upload stream, ahead of the event that frame waits for) —
        // not once per chunk (round 3: a whole-scene pass per upload call, O(chunks x N) while a scene streams in)
        if (sc->finalized && sc->block_bounds) sc->bounds_dirty = true;
        HIP_TRY(hipEventRecord(sc->upload_done, us));
        sc->any_upload = true;
    }
    return GSPLAT_OK;
}

// ... and the frame's view of the scene's arrays: taken under the scene's lock TOGETHER with the wait on upload_done, so a
// frame either sees no gather slots (and renders without them) or slots whose fill it is ordered behind — never a pointer
// another thread's first higher-band upload published a moment ago (round 4 read sc->soa unlocked in render_front)
int wait_for_uploads(gsplat_ctx *c, hipStream_t s, SceneSoA *soa_out = nullptr) {
    SceneStore *sc = c->scene.get();
    std::lock_guard<std::mutex> lock(sc->mutex);
    if (soa_out) *soa_out = sc->soa;
    if (sc->bounds_dirty) {
        // (a chunk that is enqueued on the upload stream after this pass sets the flag again when its call ends; until
        // then its splats may be missing from a culled block for a frame — like a chunk that arrives a frame later)
        launch_block_bounds(sc->soa, sc->n, sc->block_bounds, sc->upload_stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sc->upload_done, sc->upload_stream));
        sc->bounds_dirty = false;
    }
    if (sc->any_upload) HIP_TRY(hipStreamWaitEvent(s, sc->upload_done, 0));
    return GSPLAT_OK;
}

void release_async(gsplat_ctx *c) {
    gsplat_ctx::AsyncRing &a = c->async;
    if (a.stream) { (void)hipStreamSynchronize(a.stream); (void)hipStreamDestroy(a.stream); }
    for (float4 *dimg : a.dev)
        if (dimg) dev_release(c, dimg, (size_t)c->width * c->height * sizeof(float4));
    for (float *drgb : a.dev_rgb)
        if (drgb) dev_release(c, drgb, (size_t)c->width * c->height * 3 * sizeof(float));
    for (float *&h : a.host) { if (h) (void)hipHostFree(h); h = nullptr; }
    for (hipEvent_t &e : a.rendered) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t &e : a.copy_start) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t &e : a.copy_done) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (c->last_image != c->image && c->last_image != c->ext_image) c->last_image = nullptr;  // (it was a ring image)
    a = gsplat_ctx::AsyncRing();
}

void unbind_external(gsplat_ctx *c) {
    if (c->ext_mem) (void)hipDestroyExternalMemory(c->ext_mem);
    if (c->last_image == c->ext_image) c->last_image = nullptr;  // (the mapping is gone: the image tap must not read it)
    c->ext_mem = nullptr;
    c->ext_image = nullptr;
}

float4 *default_target(gsplat_ctx *c) { return c->ext_image ? c->ext_image : c->image; }

void forget_history(gsplat_ctx *c) {
    c->front_done = false;
    c->front_batch.count = 0u;
    c->last_batch = 0u;
    c->rendered = false;
    c->bigs_unknown = 3;      // (whether this context's emissions meet big rectangles has to be learnt again)
    c->last_image = nullptr;  // no frame of this context's current state exists: the image tap falls back to c->image
}

// words of one pair-key buffer (+ 16 bytes: whole-vector loads of a last, partial lane)
size_t key_words(uint64_t capacity, bool wide) { return (size_t)(wide ? capacity : (capacity + 1) / 2) + 4; }

// RasterizeData[N] on demand (zeroed: the records of splats a frame does not write must read as "no record")
size_t record_slots(const gsplat_ctx *c) { return c->batch > 1 ? (size_t)c->batch * c->n_pad : (size_t)c->n; }
int ensure_culled(gsplat_ctx *c) {
    if (c->culled) return GSPLAT_OK;
    return dev_alloc(c, &c->culled, record_slots(c) * 3, true);
}
// ... and the staged geometry of geometry-eager lazy frames (32 B per slot; the compositor only reads slots its tile lists
// name, and those were written by the same frame's projection kernel)
int ensure_geo(gsplat_ctx *c) {
    if (c->geo) return GSPLAT_OK;
    return dev_alloc(c, &c->geo, record_slots(c) * 2, false);
}

// 32-bit pair keys from now on (gsplat_finalize_scene; the Morton sort itself): the 16-bit buffers are replaced
int ensure_wide_keys(gsplat_ctx *c) {
    if (c->keys_wide) return GSPLAT_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    // the wide buffers first, the narrow ones go only once both exist: a failed allocation leaves the context as it was
    // (round 4 freed first: an out-of-memory here left null key buffers behind, and the next frame faulted on the GPU)
    uint32_t *wide[2] = {nullptr, nullptr};
    for (int h = 0; h < 2; ++h) {
        const int rc = dev_alloc(c, &wide[h], key_words(c->capacity, true), false);
        if (rc != GSPLAT_OK) {
            for (int k = 0; k < h; ++k) dev_release(c, wide[k], key_words(c->capacity, true) * sizeof(uint32_t));
            return rc;
        }
    }
    for (int h = 0; h < 2; ++h) {
        dev_release(c, c->sort.keys[h], key_words(c->capacity, false) * sizeof(uint32_t));
        c->sort.keys[h] = wide[h];
    }
    c->keys_wide = true;
    forget_history(c);  // (the last frame's sorted keys went with the old buffers)
    return GSPLAT_OK;
}

// The count matrix of the one-pass pair sort (sort.hip "wide" pass), once per context: sized for the largest bin count any
// stripe of this frame can ask for (1024 bins = 4 MiB up to 1024 tiles in all, 4096 bins = 16 MiB above), so that
// gsplat_set_stripe / gsplat_group_set_cuts never allocate and no frame ever does (round 5 allocated on first use inside
// render_front: a hipFree / hipMalloc on the frame path stalls every other context on the device).  Contexts that cannot
// take the form — 32-bit keys, at most 256 tiles, ballot ranking, GSPLAT_PAIR_SORT=split — hold none.
int ensure_wide_hist(gsplat_ctx *c) {
    if (c->keys_wide || c->pair_sort_policy == 1 || !c->sort.rank_atomic) return GSPLAT_OK;
    const uint32_t tiles = c->gx * c->gy * c->batch;
    const uint32_t bins = tiles <= 256u ? 0u : (tiles <= 1024u ? 1024u : 4096u);
    if (bins <= c->sort.wide_bins_allocated) return GSPLAT_OK;
    uint32_t *hist = nullptr;
    const int rc = dev_alloc(c, &hist, sort_wide_hist_words(bins), false);
    if (rc != GSPLAT_OK) return rc;
    if (c->sort.wide_hist) dev_release(c, c->sort.wide_hist, sort_wide_hist_words(c->sort.wide_bins_allocated) * sizeof(uint32_t));
    c->sort.wide_hist = hist;
    c->sort.wide_bins_allocated = bins;
    return GSPLAT_OK;
}

// sort_rank_selftest() once per DEVICE (not per process: the members of a one-process group sit on different devices,
// and a property of the LDS unit is a property of the chip it was measured on).  Runs on the current device = `device`.
bool rank_selftest_on(int device) {
    static std::mutex mutex;
    static int8_t known[64];  // 0 unknown, 1 lane-ordered, 2 not (or the test could not run)
    std::lock_guard<std::mutex> lock(mutex);
    if (device < 0 || device >= 64) return sort_rank_selftest();
    if (known[device] == 0) known[device] = sort_rank_selftest() ? 1 : 2;
    return known[device] == 1;
}

int ctx_create(const gsplat_config *config, std::shared_ptr<SceneStore> scene, int device, gsplat_ctx **out_ctx,
               uint32_t batch = 1) {
    const uint32_t gx = (config->width + TILE - 1) / TILE, gy = (config->height + TILE - 1) / TILE;
    const uint32_t factor = config->key_budget_factor ? config->key_budget_factor : 10u;
    const uint32_t n_cfg = scene ? scene->n : config->max_splats;
    const uint64_t capacity = (uint64_t)factor * n_cfg * batch;   // (a batch context sorts the pairs of `batch` frames at once)

    gsplat_ctx *c = new (std::nothrow) gsplat_ctx();
    if (!c) return GSPLAT_ERR_OUT_OF_MEMORY;
    c->cfg = *config;
    c->cfg.key_budget_factor = factor;
    c->cfg.max_splats = n_cfg;
    c->device = device;
    c->n = n_cfg;
    c->batch = batch;
    c->n_pad = (uint32_t)((((size_t)n_cfg + PROJ_BLOCK - 1) / PROJ_BLOCK) * PROJ_BLOCK);
    c->capacity = capacity;
    c->width = config->width; c->height = config->height;
    c->gx = gx; c->gy = gy;

    int rc = GSPLAT_OK;
    do {
        if (!scene && (rc = scene_create(device, n_cfg, &scene))) break;
        c->scene = scene;
        if (config->stream) {
            c->stream = static_cast<hipStream_t>(config->stream);
        } else {
            hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
            if (e != hipSuccess) { rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); break; }
            c->own_stream = true;
        }
        hipError_t e;
        if ((rc = apply_stripe(c, config->stripe_axis, config->stripe_begin, config->stripe_end))) break;

        // per-splat and per-workgroup arrays: one entry per (virtual) slot / projection workgroup
        const size_t n = batch > 1 ? (size_t)batch * c->n_pad : (size_t)c->n;
        const size_t nb = (size_t)batch * scene->num_proj_blocks;
        // (RasterizeData[N], gaussian_splatting_rasterizer.gd:85: ensure_culled, on demand)
        {   // pair keys: 32-bit (tile << 16 | depth16, gsplat_projection.glsl:222) where the tie repair of a re-laid-out
            // scene compares whole keys, 16-bit tile ids otherwise (sort.hip) — and the buffers are sized for what they hold
            const char *kp = getenv("GSPLAT_KEYS");
            if (kp && !strcmp(kp, "wide")) c->wide_keys_only = true;
            c->ties_storage = (config->flags & GSPLAT_FLAG_TIES_STORAGE_ORDER) != 0;
            c->keys_wide = c->wide_keys_only || (scene->finalized && !c->ties_storage);
        }
        if ((rc = dev_alloc(c, &c->keys.key, n, true))) break;
        if ((rc = dev_alloc(c, &c->keys.dims, n, true))) break;
        if ((rc = dev_alloc(c, &c->block_sums, nb, true))) break;
        if ((rc = dev_alloc(c, &c->emit_sums, nb, true))) break;
        if ((rc = dev_alloc(c, &c->block_base, nb, true))) break;
        if ((rc = dev_alloc(c, &c->block_skip, nb, true))) break;
        if ((rc = dev_alloc(c, &c->live_lists, live_list_words(nb), true))) break;
        if ((rc = dev_alloc(c, &c->big_list, (size_t)emit_big_list_entries(capacity) * 2, false))) break;
        c->long_capacity = (uint32_t)(capacity / 65u) + 2u;
        if ((rc = dev_alloc(c, &c->long_list, (size_t)c->long_capacity, false))) break;
        for (int h = 0; h < 2 && !rc; ++h) {  // gaussian_splatting_rasterizer.gd:87-89: ping-pong halves
            if ((rc = dev_alloc(c, &c->sort.keys[h], key_words(capacity, c->keys_wide), false))) break;
            if ((rc = dev_alloc(c, &c->sort.values[h], (size_t)capacity, false))) break;
            if ((rc = dev_alloc(c, &c->sort.list[h].key, n, false))) break;
            if ((rc = dev_alloc(c, &c->sort.list[h].id, n, false))) break;
            if ((rc = dev_alloc(c, &c->sort.list[h].dims, n, false))) break;
        }
        if (rc) break;
        if (config->flags & GSPLAT_FLAG_KEEP_EMITTED) {
            if ((rc = dev_alloc(c, &c->emit_keys, (size_t)capacity, false))) break;
            if ((rc = dev_alloc(c, &c->emit_values, (size_t)capacity, false))) break;
        }
        if ((rc = dev_alloc(c, &c->sort.part_hist, (size_t)sort_max_partitions(capacity) * 256, true))) break;
        if ((rc = dev_alloc(c, &c->sort.splat_hist, nb * 256, true))) break;
        if ((rc = dev_alloc(c, &c->sort.digit_base, 256, true))) break;
        {
            const char *cp = getenv("GSPLAT_COLOR");  // lazy | eager: pin where the SH colours are evaluated (A/B, tests)
            c->color_policy = cp && (!strcmp(cp, "lazy") || !strcmp(cp, "compositor")) ? 1
                              : (cp && (!strcmp(cp, "eager") || !strcmp(cp, "all")) ? 2 : 0);
            // three words the scan kernel posts to the host every frame (no copy, no synchronisation): the host reads
            // whatever is there when it sets up the next frame
#ifdef GSPLAT_TEST_HOOKS
            if (const char *pc = getenv("GSPLAT_PROBE_RENDER_CUS")) {
                // groups of 8 mask bits (single alternating bits restrict nothing on this device, round 5): of every 8
                // groups, the first cus / 32 stay
                const int keep = atoi(pc) / 32;
                uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int bit = 0; bit < 256; ++bit)
                    if (((bit >> 3) & 7) < keep) mask[bit >> 5] |= 1u << (bit & 31);
                if (keep >= 1 && keep <= 8 &&
                    hipExtStreamCreateWithCUMask(&c->probe_render_stream, 8, mask) == hipSuccess) {
                    (void)hipEventCreateWithFlags(&c->probe_render_ready, hipEventDisableTiming);
                    (void)hipEventCreateWithFlags(&c->probe_render_done, hipEventDisableTiming);
                } else {
                    c->probe_render_stream = nullptr;
                    (void)hipGetLastError();
                }
            }
#endif
            const char *lp = getenv("GSPLAT_LIVE_LISTS");
            if (lp && (!strcmp(lp, "on") || !strcmp(lp, "1"))) c->use_live_lists = true;
            else if (lp && (!strcmp(lp, "off") || !strcmp(lp, "0"))) c->use_live_lists = false;
            const char *gp = getenv("GSPLAT_GEO");    // on | off: who produces the staged geometry in lazy frames (A/B, tests)
            if (gp && (!strcmp(gp, "on") || !strcmp(gp, "1"))) c->geo_policy = 1;
            else if (gp && (!strcmp(gp, "off") || !strcmp(gp, "0"))) c->geo_policy = 0;
            hipError_t he = hipHostMalloc(reinterpret_cast<void **>(&c->hint_host), 64, hipHostMallocMapped);
            if (he != hipSuccess) { rc = hip_fail(he, "hipHostMalloc", __FILE__, __LINE__); break; }
            memset(c->hint_host, 0, 64);
            he = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->hint_dev), c->hint_host, 0);
            if (he != hipSuccess) { rc = hip_fail(he, "hipHostGetDevicePointer", __FILE__, __LINE__); break; }
            const char *rp = getenv("GSPLAT_ROUNDS");  // off | <fraction of the visible splats in round A> (A/B, tests)
            if (rp && (!strcmp(rp, "off") || !strcmp(rp, "1"))) c->rounds_policy = 1;
            else if (rp && atof(rp) > 0.0 && atof(rp) < 1.0) {
                c->rounds_policy = 2;
                c->rounds_frac16 = (uint32_t)(atof(rp) * 65536.0);
                if (c->rounds_frac16 == 0u) c->rounds_frac16 = 1u;
            }
            c->rounds_ctl.debug = getenv("GSPLAT_DEBUG_ROUNDS") != nullptr;
            c->rounds_ctl.tag = c;
            const char *op = getenv("GSPLAT_TILE_ORDER");
            if (op && !strcmp(op, "rows")) c->order_mode = ORDER_ROWS;
            else if (op && !strcmp(op, "lpt")) c->order_mode = ORDER_LPT;
            else if (op && !strcmp(op, "xcd")) c->order_mode = ORDER_XCD;
            const char *ps = getenv("GSPLAT_PAIR_SORT");
            c->pair_sort_policy = ps && !strcmp(ps, "split") ? 1 : (ps && !strcmp(ps, "wide") ? 2 : 0);
            const char *sp = getenv("GSPLAT_SORT_SMALL");  // A/B and tests: 0 = never 1024-element partitions
            c->sort.small_count = sp ? (uint32_t)strtoul(sp, nullptr, 10) : sort_small_count_default();
            if (c->sort.small_count > sort_small_count_default()) c->sort.small_count = sort_small_count_default();
            // ranking inside the sort's downsweeps: returning LDS atomics where the device hands them out in lane
            // order (checked once per process), the ballot form otherwise or on request
            const char *rk = getenv("GSPLAT_SORT_RANK");
            if (rk && !strcmp(rk, "ballot")) c->sort.rank_atomic = false;
            else c->sort.rank_atomic = rank_selftest_on(device);
        }
        if ((rc = ensure_wide_hist(c))) break;
        if ((rc = dev_alloc(c, &c->pick, 1, true))) break;
        if ((rc = dev_alloc(c, &c->counters, 1, true))) break;
        c->sort.v_count = &c->counters->v_count;
        SizeBuffers sb;
        if ((rc = alloc_size_dependent(c, c->width, c->height, gx, gy, &sb))) break;
        c->bounds = sb.bounds; c->tile_staged = sb.tile_staged; c->tile_order = sb.tile_order; c->image = sb.image;
        c->tile_done = sb.tile_done; c->tile_sat = sb.tile_sat; c->edge_t = sb.edge_t;
        for (int i = 0; i < 7 && !rc; ++i) {
            e = hipEventCreate(&c->ev[i]);
            if (e != hipSuccess) rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
            if (rc == GSPLAT_OK && i < 4) {
                if (hipEventCreate(&c->rounds_ring[i].start) != hipSuccess || hipEventCreate(&c->rounds_ring[i].end) != hipSuccess)
                    rc = hip_fail(hipErrorUnknown, "hipEventCreate", __FILE__, __LINE__);
            }
        }
        if (rc) break;
        if (config->flags & GSPLAT_FLAG_KERNEL_TIMING) {
            for (int i = 0; i <= KernelTimer::MAX_MARKS && !rc; ++i) {
                e = hipEventCreate(&c->kt.ev[i]);
                if (e != hipSuccess) rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
            }
            if (rc) break;
            c->kt.enabled = true;
            c->kt_events_created = true;
        }
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { rc = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__); break; }
        std::lock_guard<std::mutex> lock(scene->mutex);
        scene->views.push_back(c);
    } while (0);
    if (rc != GSPLAT_OK) {
        gsplat_destroy(c);
        return rc;
    }
    *out_ctx = c;
    return GSPLAT_OK;
}

int check_config(const gsplat_config *config) {
    if (config->struct_size != sizeof(gsplat_config)) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (config->width == 0 || config->height == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    const uint32_t gx = (config->width + TILE - 1) / TILE, gy = (config->height + TILE - 1) / TILE;
    // 16-bit tile ids (gsplat_projection.glsl:222) and 16-bit rectangle sizes
    if ((uint64_t)gx * gy > 65536ull || gx > 65535u || gy > 65535u) return GSPLAT_ERR_OUT_OF_RANGE;
    if (config->sh_degree < -1 || config->sh_degree > 3) return GSPLAT_ERR_INVALID_ARGUMENT;
    return GSPLAT_OK;
}

}  // namespace

namespace gsplat {
CtxView ctx_view(gsplat_ctx *c) {
    const SceneStore *sc = c->scene.get();
    return CtxView{c->device, c->stream, c->width, c->height, c->gx, c->gy, default_target(c),
                   (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0,
                   (c->cfg.flags & GSPLAT_FLAG_BLOCK_CULL) != 0 && sc->finalized && sc->block_bounds != nullptr,
                   c->ties_storage};
}
void ctx_record_gather(gsplat_ctx *c, hipEvent_t start, hipEvent_t stop) { c->gather_start = start; c->gather_stop = stop; }
void ctx_set_last_image(gsplat_ctx *c, float4 *image) { c->last_image = image; }
bool ctx_join_group(gsplat_ctx *c, const void *group) {
    if (group != nullptr && c->group != nullptr) return false;  // one group at a time
    c->group = group;
    return true;
}
int set_last_error(const char *text, int status) {
    snprintf(g_last_error, sizeof g_last_error, "%s", text);
    return status;
}
}  // namespace gsplat

extern "C" {

int gsplat_create(const gsplat_config *config, gsplat_ctx **out_ctx) {
    if (!config || !out_ctx) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int rc = check_config(config);
    if (rc) return rc;
    const uint32_t factor = config->key_budget_factor ? config->key_budget_factor : 10u;
    if ((uint64_t)factor * config->max_splats >= 0xFFFFF000ull) return GSPLAT_ERR_OUT_OF_RANGE;  // pair indices are 32-bit
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return GSPLAT_ERR_NO_DEVICE;
    }
    int device = config->device_id;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return GSPLAT_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    return ctx_create(config, nullptr, device, out_ctx);
}

int gsplat_create_view(gsplat_ctx *owner, const gsplat_config *config, gsplat_ctx **out_ctx) {
    if (!owner || !config || !out_ctx) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    int rc = check_config(config);
    if (rc) return rc;
    if (config->max_splats != 0 && config->max_splats != owner->n) return GSPLAT_ERR_INVALID_ARGUMENT;
    const uint32_t factor = config->key_budget_factor ? config->key_budget_factor : 10u;
    if ((uint64_t)factor * owner->n >= 0xFFFFF000ull) return GSPLAT_ERR_OUT_OF_RANGE;
    HIP_TRY(hipSetDevice(owner->device));
    return ctx_create(config, owner->scene, owner->device, out_ctx);
}

int gsplat_destroy(gsplat_ctx *c) {
    if (!c) return GSPLAT_OK;
    if (c->group)
        return set_last_error("gsplat_destroy: the context is a member of a gsplat_group; destroy the group first",
                              GSPLAT_ERR_INVALID_ARGUMENT);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    release_async(c);
    unbind_external(c);
    if (c->scene) {
        std::lock_guard<std::mutex> lock(c->scene->mutex);
        auto &v = c->scene->views;
        v.erase(std::remove(v.begin(), v.end(), c), v.end());
    }
    for (void *p : c->allocations) (void)hipFree(p);
    if (c->hint_host) (void)hipHostFree(c->hint_host);
    for (int i = 0; i < 7; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (gsplat_ctx::RoundsSlot &sl : c->rounds_ring) {
        if (sl.start) (void)hipEventDestroy(sl.start);
        if (sl.end) (void)hipEventDestroy(sl.end);
    }
    if (c->kt_events_created)
        for (int i = 0; i <= KernelTimer::MAX_MARKS; ++i) (void)hipEventDestroy(c->kt.ev[i]);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    c->scene.reset();  // the last context on a scene frees it
    delete c;
    return GSPLAT_OK;
}

int gsplat_upload_splats(gsplat_ctx *c, uint32_t first, uint32_t count, const float *records60) {
    return upload_common(c, first, count, records60, GSPLAT_RECORD_FLOATS, false, 0.0f);
}

int gsplat_upload_ply_rows(gsplat_ctx *c, uint32_t first, uint32_t count, const float *rows62, float load_time) {
    return upload_common(c, first, count, rows62, GSPLAT_PLY_ROW_FLOATS, true, load_time);
}

int gsplat_finalize_scene(gsplat_ctx *c) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    SceneStore *sc = c->scene.get();
    if (sc->finalized || sc->n < 2) return GSPLAT_OK;
    HIP_TRY(hipSetDevice(sc->device));
    std::lock_guard<std::mutex> lock(sc->mutex);
    HIP_TRY(hipStreamSynchronize(sc->upload_stream));
    for (gsplat_ctx *v : sc->views) {  // no frame of any context may be reading the scene while it is permuted
        HIP_TRY(hipStreamSynchronize(v->stream));
        v->front_done = false;  // a frame begun on the old layout cannot be ended on the new one
    }
    const uint32_t n = sc->n;
    hipStream_t s = sc->upload_stream;
    int rc;
    // the pair-level buffers of every context on the scene carry 32-bit keys from here on (the tie repair of a re-laid-out
    // scene compares whole keys; the Morton sort below borrows this context's as N 30-bit codes)
    // (views that keep equal keys in storage order — GSPLAT_FLAG_TIES_STORAGE_ORDER — have no repair pass and keep 16-bit keys)
    for (gsplat_ctx *v : sc->views)
        if (!v->ties_storage && (rc = ensure_wide_keys(v)) != GSPLAT_OK) return rc;
    // 30-bit Morton code of the position inside the bounding box of the finite positions, and the stable order of
    // (code, id): on the device — two small kernels and the context's own pair sort (four 8-bit passes over N (code,
    // id) pairs in its sort buffers; every stream of the scene is idle here).  Round 2 did this on the host (a copy of
    // all positions and a std::sort of N words: seconds at 30 M splats).
    if (!sc->id_of_slot) {
        if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->id_of_slot, (size_t)n, false, s))) return rc;
        if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->slot_of_id, (size_t)n, false, s))) return rc;
    }
    {
        uint32_t *box6 = c->sort.digit_base;  // (256 words of per-pass scratch: free until the sort below starts)
        // N 32-bit codes: this context's own key buffers where they hold that many words, scratch otherwise (16-bit key
        // buffers of a context with a small key budget)
        SortBuffers sb = c->sort;
        struct Scratch {
            uint32_t *p[2] = {nullptr, nullptr};
            ~Scratch() { for (uint32_t *q : p) if (q) (void)hipFree(q); }
        } scratch;
        if (key_words(c->capacity, c->keys_wide) < (size_t)n + 4)
            for (int h = 0; h < 2; ++h) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&scratch.p[h]), ((size_t)n + 4) * sizeof(uint32_t)));
                sb.keys[h] = scratch.p[h];
            }
        launch_morton_keys(sc->soa.pos_time, n, box6, sb.keys[0], sb.values[0], s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&c->counters->d_sorted), (int)n, 1, s));
        const int half = launch_sort_pairs(sb, &c->counters->d_sorted, c->capacity, 30, s, nullptr, 0, false);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(sc->id_of_slot, sb.values[half], (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        launch_invert_permutation(sc->id_of_slot, n, sc->slot_of_id, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));  // (the scratch goes at the end of this block)
    }
    // permute the scene arrays through one temporary (the largest: 12 float4 of SH coefficients per splat)
    float4 *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)n * (sc->soa.sh_block ? SH_BLOCK_F4 : 1) * sizeof(float4)));
    struct Arr { float4 *arr; uint32_t rec; };
    const Arr arrays[] = {{sc->soa.pos_time, 1u}, {sc->soa.cov_a, 1u}, {sc->soa.cov_b, 1u}, {sc->soa.sh_dc, 1u},
                          {sc->soa.sh_block, (uint32_t)SH_BLOCK_F4}};
    for (const auto &a : arrays) {
        if (a.arr == nullptr) continue;  // (a band-0 scene has no slots)
        launch_permute_float4(a.arr, tmp, sc->id_of_slot, n, a.rec, s);
        hipError_t e = hipMemcpyAsync(a.arr, tmp, (size_t)n * a.rec * sizeof(float4), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { (void)hipFree(tmp); return hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__); }
    }
    hipError_t e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "scene re-layout", __FILE__, __LINE__);
    if (!sc->block_bounds)
        if ((rc = raw_alloc(sc->allocations, sc->bytes, &sc->block_bounds, (size_t)sc->num_proj_blocks * 3, false, s)))
            return rc;
    launch_block_bounds(sc->soa, n, sc->block_bounds, s);  // on the upload stream, complete before this call returns
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    sc->finalized = true;
    for (gsplat_ctx *v : sc->views) forget_history(v);  // marks and taps were indexed by the old slots
    return GSPLAT_OK;
}

int gsplat_resize(gsplat_ctx *c, uint32_t width, uint32_t height) {
    if (!c || width == 0 || height == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    const uint32_t gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    if ((uint64_t)gx * gy > 65536ull || gx > 65535u || gy > 65535u) return GSPLAT_ERR_OUT_OF_RANGE;
    if (c->group)  // (the group's stripes, staging buffers and broadcast counts are sized for the current frame)
        return set_last_error("gsplat_resize: the context is a member of a gsplat_group; destroy the group, resize every "
                              "member, create the group again", GSPLAT_ERR_INVALID_ARGUMENT);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // gaussian_splatting_rasterizer.gd:26-48: new tile_bounds and image.  The new buffers are allocated before the
    // old ones go, so a failure leaves the context as it was; a frame begun with gsplat_render_begin is dropped.
    SizeBuffers nb;
    int rc = alloc_size_dependent(c, width, height, gx, gy, &nb);
    if (rc != GSPLAT_OK) {
        release_size_dependent(c, nb, width, height, gx, gy);
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    release_async(c);    // (sized for the old frame; re-made by the next gsplat_render_async)
    unbind_external(c);  // the imported image had the old size: the host binds the new texture
    const SizeBuffers old{c->bounds, c->tile_staged, c->tile_order, c->tile_done, c->tile_sat, c->edge_t, c->image};
    release_size_dependent(c, old, c->width, c->height, c->gx, c->gy);
    c->bounds = nb.bounds; c->tile_staged = nb.tile_staged; c->tile_order = nb.tile_order; c->image = nb.image;
    c->tile_done = nb.tile_done; c->tile_sat = nb.tile_sat; c->edge_t = nb.edge_t;
    c->width = width; c->height = height; c->gx = gx; c->gy = gy;
    c->cfg.width = width; c->cfg.height = height;
    // a stripe is expressed in tiles of the old grid: fall back to the full frame
    (void)apply_stripe(c, GSPLAT_STRIPE_NONE, 0, 0);
    forget_history(c);
    if (ensure_wide_hist(c) != GSPLAT_OK) (void)hipGetLastError();  // (a larger grid may allow more bins; without them: split passes)
    return GSPLAT_OK;
}

int gsplat_set_stripe(gsplat_ctx *c, uint32_t axis, uint32_t b, uint32_t e) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    const int rc = apply_stripe(c, axis, b, e);
    if (rc != GSPLAT_OK) return rc;
    forget_history(c);  // the last frame's taps / pick / begun frame belong to the old stripe
    // ... and so do the per-tile staged counts the colour policy sums up (tiles outside the new stripe would keep theirs)
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(c->tile_staged, 0, (size_t)c->gx * c->gy * c->batch * sizeof(uint32_t), c->stream));
    return GSPLAT_OK;
}

static bool is_sharded(const gsplat_ctx *c) {
    return c->sx0 > 0 || c->sy0 > 0 || c->sx1 < c->gx || c->sy1 < c->gy;
}

// First half of a frame: projection, splat sort, key emission, pair sort.  stripe_cull: workgroups that cannot reach
// the context's stripe may be skipped too — then the "last tile" counter is stripe-local and the caller of render_back
// supplies the frame's (gsplat_render_end); without it only workgroups outside a frustum plane are skipped.
// Does this frame run in two rounds, and how large is round A?  Two rounds need: no heat map and no pick in the frame
// (both read a tile's TOTAL pair count), no emission-order tap, at most 32 768 tiles.  Whether they pay is measured, not
// guessed: rounds_controller.h decides from the frame times collected here.
static bool choose_rounds(gsplat_ctx *c, const gsplat_frame *frame, uint32_t tiles) {
    c->rounds_slot = -1;
    if (c->rounds_policy == 1 || tiles > ROUNDS_MAX_TILES) return false;
    if (frame->heatmap_factor != 0.0f || frame->target_tile != GSPLAT_NO_TARGET_TILE) return false;
    if (c->cfg.flags & GSPLAT_FLAG_KEEP_EMITTED) return false;
    if (c->rounds_policy == 2) return true;  // pinned fraction
    RoundsController &ctl = c->rounds_ctl;
    // hand over the frame times that have become available
    for (gsplat_ctx::RoundsSlot &sl : c->rounds_ring) {
        if (!sl.pending || hipEventQuery(sl.end) != hipSuccess) continue;
        sl.pending = false;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sl.start, sl.end) == hipSuccess) ctl.observe(sl.trial, sl.counts, ms);
    }
    bool wants_timing = false, counts = false;
    const bool two = ctl.begin_frame(&wants_timing, &counts);
    c->rounds_frac16 = ctl.frac16;
    // time this frame if a ring slot is free
    gsplat_ctx::RoundsSlot &sl = c->rounds_ring[c->rounds_next_slot];
    if (wants_timing && !sl.pending && sl.start != nullptr) {
        sl.trial = ctl.trial;
        sl.counts = counts;
        c->rounds_slot = c->rounds_next_slot;
        c->rounds_next_slot = (c->rounds_next_slot + 1) & 3;
    }
    return two;
}

// The sorted pairs, tile_bounds and the pick of a two-round frame: the frame once more, in one round, without the
// compositor (the image, the staged counts and the statistics of the frame stay as they are).
static int render_front(gsplat_ctx *c, const gsplat_frame *frame, bool stripe_cull, bool replay = false,
                        uint32_t *last_tile_copy = nullptr);
static int render_back(gsplat_ctx *c, float4 *target, uint32_t pitch, uint32_t ox, uint32_t oy,
                       const uint32_t *last_tile_dev, bool no_render = false);
static int replay_full(gsplat_ctx *c) {
    if (!c->rendered || !c->taps_stale) return GSPLAT_OK;
    const gsplat_frame frame = c->last_frame;
    int rc = render_front(c, &frame, c->last_stripe_cull, /*replay=*/true);
    if (rc != GSPLAT_OK) return rc;
    rc = render_back(c, nullptr, 0, 0, 0, &c->counters->replay_last_tile_plus1, /*no_render=*/true);
    if (rc != GSPLAT_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSPLAT_OK;
}

// replay: the last frame once more in one round, for its taps (no timing, no hint postings, same colour mode).
static int render_front(gsplat_ctx *c, const gsplat_frame *frame, bool stripe_cull, bool replay, uint32_t *last_tile_copy) {
    hipStream_t s = c->stream;
    SceneStore *sc = c->scene.get();
    FrameParams fp;
    fill_frame_params(c, frame, &fp);
    const bool timing = !replay && (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0;
    const int sh_degree = c->cfg.sh_degree >= 0 ? c->cfg.sh_degree : sc->sh_degree_seen.load();
    const uint32_t tiles = c->gx * c->gy;
    // key bits the pair level sorts: 16-bit keys are stripe-local tile ids (TileMap), 32-bit keys carry the frame's
    const bool narrow = !c->keys_wide;  // (a frame has at most 65 536 tiles: gsplat_create; wide: GSPLAT_KEYS=wide, finalized scenes)
    const uint32_t stripe_tiles = (c->sx1 - c->sx0) * (c->sy1 - c->sy0);
    const int sig_bits = sig_bits_for(narrow ? (stripe_tiles ? stripe_tiles : 1u) : tiles);
    KernelTimer *kt = (c->kt.enabled && !replay) ? &c->kt : nullptr;
    c->front_done = false;
    c->front_batch.count = 0u;
    SceneSoA soa;  // (the degree was read BEFORE this snapshot: an upload raises it only after its slots exist)
    int rc = wait_for_uploads(c, s, &soa);
    if (rc) return rc;
    const bool rounds = !replay && choose_rounds(c, frame, tiles);
    uint32_t *hints = replay ? nullptr : c->hint_dev;
    if (!replay && c->rounds_slot >= 0) HIP_TRY(hipEventRecord(c->rounds_ring[c->rounds_slot].start, s));

    // Who evaluates the SH colours (gsplat_projection.glsl:198-201)?  Eager = the projection kernel, for all V visible
    // splats (band-0 scenes: 16 B streamed per splat; higher bands: the splat's 192-byte coefficient block, every lane
    // its own — measured 59 ns per splat-colour); lazy = the compositor, for the D_c pairs it stages, gathering the same
    // block (measured ~20 ns per pair: the gather hides behind the blend loops of the other tiles).  So lazy pays up
    // to D_c ~ 2.5 V, i.e. always short of close-ups where every splat is composited in several tiles: 6 M splats at
    // 1080p (D_c = 3.0 M, V = 5.9 M) +19 % fps, the 4K config +12 %.  V and D_c of the previous frames come from the
    // words the scan kernel posts to host memory; hysteresis 2.25 / 2.75; no history yet: lazy.
    bool lazy = c->last_lazy;
    if (replay) {
        // (the records of the frame being replayed were written in this mode)
    } else if (sh_degree <= 0 || c->color_policy == 2) {
        lazy = false;  // band 0 only: 16 bytes per splat are cheaper to stream than to gather
    } else if (c->color_policy == 1) {
        lazy = true;
    } else {
        const volatile uint32_t *h = c->hint_host;
        const uint32_t v_prev = h[0], dc_prev = h[1], frames = h[2];
        if (frames < 2u) lazy = true;
        else if ((uint64_t)dc_prev * 4u < (uint64_t)v_prev * 9u) lazy = true;
        else if ((uint64_t)dc_prev * 4u > (uint64_t)v_prev * 11u) lazy = false;
    }
    c->front_lazy = lazy;
    const bool geo = lazy && sh_degree > 0 && (replay ? c->last_geo : c->geo_policy == 1);
    c->front_geo = geo;
    if (geo) {
        const int grc = ensure_geo(c);
        if (grc != GSPLAT_OK) return grc;
    }
    if (sh_degree > 0 && soa.sh_block == nullptr) {  // (bands forced by gsplat_config.sh_degree on a band-0 scene)
        std::lock_guard<std::mutex> lock(sc->mutex);
        const int src_ = ensure_slots(sc);
        if (src_ != GSPLAT_OK) return src_;
        HIP_TRY(hipStreamWaitEvent(s, sc->upload_done, 0));
        soa = sc->soa;
    }
    if (!lazy) {  // an eager frame writes RasterizeData
        const int erc = ensure_culled(c);
        if (erc != GSPLAT_OK) return erc;
    }

    const float4 *block_bounds = nullptr;
    if ((c->cfg.flags & GSPLAT_FLAG_BLOCK_CULL) && sc->finalized && sc->block_bounds) {
        block_bounds = sc->block_bounds;
        fp.cull_mode = stripe_cull ? 2u : 1u;
    }

    // gaussian_splatting_rasterizer.gd:127-128 clears the pair counter and tile_bounds with two buffer_clear calls;
    // here scan_blocks_kernel overwrites every per-frame counter and zeroes tile_bounds itself (no fill launches, no
    // copies: a one-round frame is 17 kernel launches and nothing else on the stream — 18 with the list of big
    // rectangles, which gets its launch only in the frames after one that met any).
    if (timing) HIP_TRY(hipEventRecord(c->ev[0], s));  // 'Start'
    if (!replay) c->kt.begin(s);
    launch_project(soa, c->n, fp, lazy ? (geo ? -2 : -1) : sh_degree, geo ? c->geo : c->culled, c->keys, c->block_sums, c->sort.splat_hist,
                   block_bounds, c->block_skip, replay ? nullptr : c->tile_staged, tiles, c->counters->dc_parts,
                   replay ? TileSchedule{} : scheduled_tiles(c, fp), s, c->use_live_lists ? c->live_lists : nullptr, sort_splat_part_blocks(c->n));
    // two-round frame: D, V and the size of round A from the projection workgroups' records (D to the host as well)
    if (rounds)
        launch_frame_plan(c->block_sums, sc->num_proj_blocks, c->capacity, c->rounds_frac16, &c->counters->total_emitted,
                          &c->counters->plan, hints ? hints + 6 : nullptr, s);
    if (kt) kt->mark(GSPLAT_KERNEL_PROJECT);
    if (timing) HIP_TRY(hipEventRecord(c->ev[1], s));
    // (with block culling the skipped workgroups wrote nothing: the sort reads their marks instead)
    const uint32_t *skip_marks = (block_bounds != nullptr && fp.cull_mode != 0u) ? c->block_skip : nullptr;
    c->front_skip_marks = skip_marks != nullptr;
    launch_sort_splats(c->sort, c->keys, c->n, skip_marks, s, kt, c->use_live_lists ? c->live_lists : nullptr);
    if (rounds && sc->finalized && !c->ties_storage)  // (a run of equal keys must not be cut where it is repaired as a whole)
        launch_plan_align(c->sort.list[0].key, c->sort.v_count, &c->counters->plan, s);
    if (timing) HIP_TRY(hipEventRecord(c->ev[2], s));
    // (round A = the first plan.v_a entries of the sorted list: the emission kernels take that word as the list length)
    const uint32_t *list_len = rounds ? &c->counters->plan.v_a : c->sort.v_count;
    launch_emit_sums(c->sort.list[0], list_len, c->n, c->emit_sums, s);
    launch_scan_blocks(c->emit_sums, c->block_sums, sc->num_proj_blocks, c->block_base, c->capacity,
                       rounds ? &c->counters->round_total[0] : &c->counters->total_emitted, &c->counters->d_sorted,
                       &c->counters->overflow, &c->counters->visible, &c->counters->frame_last_tile_plus1, c->bounds,
                       (uint32_t)bounds_entries(c->gx, c->gy), &c->counters->big_count, hints, c->counters->dc_parts,
                       hints ? hints + 4 : nullptr, last_tile_copy, &c->counters->long_count,
                       &c->counters->big_seen, s);
    if (kt) kt->mark(GSPLAT_KERNEL_SCAN);
    // rectangles of more than 512 tiles get a launch of their own (the whole grid shares each) only while this context
    // meets any: the emission counts them, the next scan posts the count to the host (hint word 3)
    // — and while the context cannot know yet (its first frames, the frames after a new stripe / size / scene layout: the
    // count reaches the host two frames late at best): unlisted, a wave that owns a screen-filling rectangle walks up to
    // 65 536 tiles by itself — the millisecond-scale stall the second launch exists to avoid; listed for nothing, the
    // launch costs a few microseconds
    const bool list_bigs = c->bigs_unknown > 0 ||
                           (c->hint_host != nullptr && reinterpret_cast<const volatile uint32_t *>(c->hint_host)[3] != 0u);
    if (!replay && c->bigs_unknown > 0) --c->bigs_unknown;
    c->front_big_hint = c->hint_host != nullptr ? reinterpret_cast<const volatile uint32_t *>(c->hint_host)[3] : 0u;
    // (a short round A = few, large splats: several workgroups per block of the list, ~16 k waves in all)
    uint32_t split = 1;
    if (rounds) {
        const uint64_t waves = ((uint64_t)c->n * c->rounds_frac16 >> 16) / 64u + 1u;
        split = (uint32_t)std::min<uint64_t>(16u, std::max<uint64_t>(1u, 16384u / waves));
    }
    launch_emit(c->sort.list[0], list_len, c->n, fp, c->emit_sums, c->block_base, c->capacity, c->sort.keys[0],
                c->sort.values[0], &c->counters->big_count, c->big_list, narrow, s, split, list_bigs, c->front_big_hint);
    c->front_list_bigs = list_bigs;
    if (kt) kt->mark(GSPLAT_KERNEL_EMIT);
    if (c->emit_keys) {
        if (narrow)
            launch_widen_keys(reinterpret_cast<const uint16_t *>(c->sort.keys[0]), c->sort.values[0], c->keys.key,
                              &c->counters->d_sorted, c->emit_keys, tile_map_of(fp), s);
        else
            HIP_TRY(hipMemcpyAsync(c->emit_keys, c->sort.keys[0], (size_t)c->capacity * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->emit_values, c->sort.values[0], (size_t)c->capacity * 4, hipMemcpyDeviceToDevice, s));
    }
    if (timing) HIP_TRY(hipEventRecord(c->ev[3], s));  // 'Projection' (emission belongs to the reference's projection pass)
    // the pairs arrive ordered by (depth16, id): only the tile bits are left to sort — in ONE pass where the stripe has few
    // enough tiles for a counting sort on the whole (stripe-local) tile id and the frame few enough pairs for that to pay
    // (sort.hip "wide" pass: three launches instead of six; it sorts any count correctly, so the pair count of the previous
    // frames — hint word 4, posted by the scan — only has to be a good guess)
    uint32_t wide_bins = 0;
    if (replay) {
        wide_bins = c->last_wide_bins;  // (same kernels as the frame being replayed; either form gives the same arrays)
    } else if (narrow && c->pair_sort_policy != 1) {
        wide_bins = sort_wide_bins(stripe_tiles);
        const uint32_t pairs_prev = c->hint_host ? reinterpret_cast<const volatile uint32_t *>(c->hint_host)[4] : 0u;
        if (c->pair_sort_policy == 0 && pairs_prev > WIDE_AUTO_PAIRS) wide_bins = 0;
    }
    // the one-pass form ranks with returning LDS atomics only (WideCounters::take has no ballot form): where the device's
    // self-test failed, or ballots were asked for (GSPLAT_SORT_RANK=ballot), the split passes run — whatever the policy or
    // the frame being replayed says.  Its count matrix was allocated with the context (ensure_wide_hist: nothing on the
    // frame path allocates, frees or synchronises the device); a context without one keeps the split passes.
    if (!c->sort.rank_atomic || c->sort.wide_bins_allocated < wide_bins) wide_bins = 0;
    c->sorted_index = wide_bins ? launch_sort_pairs_wide(c->sort, &c->counters->d_sorted, c->capacity, wide_bins, s, kt)
                                : launch_sort_pairs(c->sort, &c->counters->d_sorted, c->capacity, sig_bits, s, kt, 16, narrow);
    c->front_wide_bins = wide_bins;
    c->front_narrow = narrow;
    c->front_rounds = rounds;
    c->front_stripe_cull = stripe_cull;
    if (!replay) c->last_frame = *frame;
    if (timing) HIP_TRY(hipEventRecord(c->ev[4], s));  // 'Sort'
    HIP_TRY(hipGetLastError());
    c->front_fp = fp;
    c->front_soa = soa;
    c->front_sig_bits = sig_bits;
    c->front_sh_degree = sh_degree;
    c->front_done = true;
    c->rendered = false;
    return GSPLAT_OK;
}

// Second half: tile ranges + compositor.  last_tile_dev: device word holding the frame's highest populated tile + 1
// (nullptr = this context's own counter).
static int render_back(gsplat_ctx *c, float4 *target, uint32_t pitch, uint32_t ox, uint32_t oy,
                       const uint32_t *last_tile_dev, bool no_render) {
    if (!c->front_done || c->front_batch.count != 0u) return GSPLAT_ERR_INVALID_ARGUMENT;  // (a begun batch ends with gsplat_render_batch_end)
    hipStream_t s = c->stream;
    SceneStore *sc = c->scene.get();
    const FrameParams &fp = c->front_fp;
    const bool timing = !no_render && (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0;
    const uint32_t tiles = c->gx * c->gy;
    KernelTimer *kt = (c->kt.enabled && !no_render) ? &c->kt : nullptr;
    const bool fix_last = (c->cfg.flags & GSPLAT_FLAG_FIX_LAST_TILE) != 0;
    const uint32_t *last_tile = last_tile_dev ? last_tile_dev : &c->counters->frame_last_tile_plus1;
    // (the replay of a frame re-reads the word its boundaries pass saw — the caller's pointer may be gone by then: the
    // boundaries launch keeps a copy)
    uint32_t *keep = &c->counters->replay_last_tile_plus1;
    const bool fast_exp = (c->cfg.flags & GSPLAT_FLAG_FAST_EXP) != 0;
    const int lazy_degree = c->front_lazy ? c->front_sh_degree : 0;
    const bool geo = c->front_geo && lazy_degree > 0;
    const float4 *records = geo ? c->geo : c->culled;   // what the compositor's staging gathers per listed splat
    int si = c->sorted_index;
    // tile ranges of the sorted array in half `half` (+ the tie repair of a re-laid-out scene, which leaves the values
    // in the other half).  A round's array ends on the round's highest tile, while quirks Q5/Q6 belong to the FRAME's:
    // rounds ask with the "sharded" form of the test — for a whole-frame array the two forms are the same test.
    auto tile_ranges = [&](int half, bool as_shard) -> int {
        if (sc->finalized && !c->ties_storage) {
            // (long_count was zeroed by the scan that preceded this round's emission)
            launch_boundaries(c->sort.keys[half], &c->counters->d_sorted, tiles, c->bounds, fix_last, as_shard, last_tile,
                              keep, c->sort.values[half], c->sort.values[half ^ 1], sc->id_of_slot,
                              &c->counters->long_count, c->long_list, c->long_capacity, false, tile_map_of(fp), s);
            launch_tie_long_runs(c->sort.keys[half], c->sort.keys[half ^ 1], c->sort.values[half], c->sort.values[half ^ 1],
                                 &c->counters->d_sorted, sc->id_of_slot, c->n, &c->counters->long_count, c->long_list,
                                 c->long_capacity, s);
            c->values_index = half ^ 1;
        } else {
            launch_boundaries(c->sort.keys[half], &c->counters->d_sorted, tiles, c->bounds, fix_last, as_shard, last_tile,
                              keep, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, c->front_narrow, tile_map_of(fp), s);
            c->values_index = half;
        }
        return GSPLAT_OK;
    };
    {
        const int rc = tile_ranges(si, is_sharded(c) || c->front_rounds);
        if (rc != GSPLAT_OK) return rc;
    }
    if (kt) kt->mark(GSPLAT_KERNEL_BOUNDARIES);
    if (timing) HIP_TRY(hipEventRecord(c->ev[5], s));  // 'Boundaries'
    if (no_render) {
        // replay for the taps: tile_bounds and the sorted pairs are what was asked for
    } else if (!c->front_rounds) {
        hipStream_t rs = s;
#ifdef GSPLAT_TEST_HOOKS
        if (c->probe_render_stream != nullptr) {   // (diagnosis builds: the compositor on its CU-masked stream)
            HIP_TRY(hipEventRecord(c->probe_render_ready, s));
            HIP_TRY(hipStreamWaitEvent(c->probe_render_stream, c->probe_render_ready, 0));
            rs = c->probe_render_stream;
        }
#endif
        launch_render(records, c->front_soa.sh_block, lazy_degree, c->sort.values[c->values_index], c->bounds, fp, target,
                      pitch, ox, oy, c->pick, c->tile_staged, scheduled_tiles(c, fp), fast_exp, rs, 0, nullptr, nullptr, nullptr, geo);
#ifdef GSPLAT_TEST_HOOKS
        if (c->probe_render_stream != nullptr) {
            HIP_TRY(hipEventRecord(c->probe_render_done, rs));
            HIP_TRY(hipStreamWaitEvent(s, c->probe_render_done, 0));
        }
#endif
        if (kt) kt->mark(GSPLAT_KERNEL_RENDER);
    } else {
        FramePlan *plan = &c->counters->plan;
        launch_render(records, c->front_soa.sh_block, lazy_degree, c->sort.values[c->values_index], c->bounds, fp, target,
                      pitch, ox, oy, c->pick, c->tile_staged, scheduled_tiles(c, fp), fast_exp, s, 1, c->tile_done, plan, c->edge_t, geo);
        if (kt) kt->mark(GSPLAT_KERNEL_RENDER);
        // round B: the rest of the list, filtered by the tiles round A left unfinished
        if (launch_tile_sat(c->tile_done, plan, fp, c->tile_sat, s) != 0) return GSPLAT_ERR_HIP;
        launch_round_filter(c->sort.list[0], c->sort.v_count, c->n, plan, c->tile_sat, c->tile_done, fp, c->sort.list[1].key,
                            c->sort.list[1].dims, c->emit_sums, s);
        launch_scan_blocks(c->emit_sums, c->block_sums, sc->num_proj_blocks, c->block_base, c->capacity,
                           &c->counters->round_total[1], &c->counters->d_sorted, &c->counters->round_overflow,
                           &c->counters->visible, &c->counters->frame_last_tile_plus1, c->bounds,
                           (uint32_t)bounds_entries(c->gx, c->gy), &c->counters->big_count, nullptr, c->counters->dc_parts,
                           c->hint_dev ? c->hint_dev + 5 : nullptr, nullptr, &c->counters->long_count,
                           &c->counters->big_seen, s);
        if (kt) kt->mark(GSPLAT_KERNEL_SCAN);
        const SplatList rest{c->sort.list[1].key, c->sort.list[0].id, c->sort.list[1].dims};
        launch_emit(rest, c->sort.v_count, c->n, fp, c->emit_sums, c->block_base, c->capacity, c->sort.keys[0],
                    c->sort.values[0], &c->counters->big_count, c->big_list, c->front_narrow, s, 1, c->front_list_bigs,
                    c->front_big_hint);
        if (kt) kt->mark(GSPLAT_KERNEL_EMIT);
        si = c->front_wide_bins ? launch_sort_pairs_wide(c->sort, &c->counters->d_sorted, c->capacity, c->front_wide_bins, s, kt)
                                : launch_sort_pairs(c->sort, &c->counters->d_sorted, c->capacity, c->front_sig_bits, s, kt, 16, c->front_narrow);
        c->sorted_index = si;
        {
            const int rc = tile_ranges(si, true);
            if (rc != GSPLAT_OK) return rc;
        }
        if (kt) kt->mark(GSPLAT_KERNEL_BOUNDARIES);
        launch_render(records, c->front_soa.sh_block, lazy_degree, c->sort.values[c->values_index], c->bounds, fp, target,
                      pitch, ox, oy, c->pick, c->tile_staged, scheduled_tiles(c, fp), fast_exp, s, 2, c->tile_done, plan, c->edge_t, geo);
        if (kt) kt->mark(GSPLAT_KERNEL_RENDER);
    }
    if (timing) HIP_TRY(hipEventRecord(c->ev[6], s));  // 'Render' (a two-round frame: everything after round A's tile ranges)
    if (!no_render && c->rounds_slot >= 0) {
        HIP_TRY(hipEventRecord(c->rounds_ring[c->rounds_slot].end, s));
        c->rounds_ring[c->rounds_slot].pending = true;
        c->rounds_slot = -1;
    }
    HIP_TRY(hipGetLastError());
    if (!no_render) {
        c->timing_valid = timing;
        c->last_rounds = c->front_rounds;  // (a replay leaves the frame's own record alone: statistics describe the frame)
    }
    c->taps_stale = c->front_rounds;
    c->last_stripe_cull = c->front_stripe_cull;
    c->last_skip_marks = c->front_skip_marks;
    c->last_sig_bits = c->front_sig_bits;
    c->last_sh_degree = c->front_sh_degree;
    c->last_lazy = c->front_lazy;
    c->last_geo = c->front_geo;
    c->last_narrow = c->front_narrow;
    c->last_wide_bins = c->front_wide_bins;
    c->last_fp = c->front_fp;
    c->last_soa = c->front_soa;
    c->last_batch = 0;
    c->front_done = false;
    c->rendered = true;
    return GSPLAT_OK;
}

extern "C++" {
namespace gsplat {
int ctx_render_begin(gsplat_ctx *c, const gsplat_frame *frame, uint32_t *last_tile_out_device, bool stripe_cull) {
    if (!c || !frame) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    // (the caller's word is written by the launch that finalises the frame's counters: no copy of its own)
    return render_front(c, frame, stripe_cull, /*replay=*/false, last_tile_out_device);
}
}  // namespace gsplat
}  // extern "C++"

static int render_impl(gsplat_ctx *c, const gsplat_frame *frame, float4 *target, uint32_t pitch, uint32_t ox,
                       uint32_t oy) {
    // one call, no exchange: a stripe context may only skip what cannot change its "last tile" counter
    const int rc = render_front(c, frame, /*stripe_cull=*/!is_sharded(c));
    if (rc != GSPLAT_OK) return rc;
    return render_back(c, target, pitch, ox, oy, nullptr);
}


// ---- batched frames (gsplat_internal.h FrameBatch): B frames of this context through ONE launch sequence ----------
// The batch is ONE frame of a virtual image that stacks the B stripes vertically; between the projection and the compositor
// the launchers see that one frame (n = B * n_pad virtual slots, B * rows tile rows).  One round, no taps of the sort
// arrays, no pick; the frame's own statistics are the batch's totals.
static int batch_front(gsplat_ctx *c, const gsplat_frame *frames, uint32_t count, bool stripe_cull, uint32_t *last_tiles_copy) {
    if (!c || !frames || count < 1u || count > c->batch) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (c->batch < 2u) return set_last_error("gsplat_render_batch: not a batch context (gsplat_create_batch_view)", GSPLAT_ERR_INVALID_ARGUMENT);
    if (c->keys_wide)
        return set_last_error("batched frames need 16-bit pair keys: a scene in upload order, or GSPLAT_FLAG_TIES_STORAGE_ORDER on a "
                              "re-laid-out one", GSPLAT_ERR_UNSUPPORTED);
    if (c->cfg.flags & GSPLAT_FLAG_KEEP_EMITTED) return GSPLAT_ERR_UNSUPPORTED;
    hipStream_t s = c->stream;
    SceneStore *sc = c->scene.get();
    const uint32_t rows = c->sy1 - c->sy0, sw = c->sx1 - c->sx0;
    if ((uint64_t)c->gx * count * rows > 65536ull)   // (a rectangle's origin tile rides in 16 bits of the splat key)
        return set_last_error("gsplat_render_batch: batch x stripe rows x tile columns exceeds 65 536 virtual tiles", GSPLAT_ERR_OUT_OF_RANGE);
    FrameBatch &fb = c->front_batch;
    const float4 *block_bounds = nullptr;
    uint32_t cull_mode = 0u;
    if ((c->cfg.flags & GSPLAT_FLAG_BLOCK_CULL) && sc->finalized && sc->block_bounds) {
        block_bounds = sc->block_bounds;
        cull_mode = stripe_cull ? 2u : 1u;
    }
    for (uint32_t k = 0; k < (uint32_t)MAX_BATCH; ++k) {
        const gsplat_frame *f = &frames[k < count ? k : count - 1u];
        fill_frame_params(c, f, &fb.f[k]);
        fb.f[k].target_tile = GSPLAT_NO_TARGET_TILE;   // (no pick inside a batch)
        fb.f[k].cull_mode = cull_mode;
    }
    fb.count = count;
    fb.n_pad = c->n_pad;
    fb.blocks = c->n_pad / PROJ_BLOCK;
    fb.rows = rows;
    fb.image_stride_px = c->width * c->height;
    // the virtual frame: gx x (count * rows) tiles, all of them this context's
    FrameParams fpv = fb.f[0];
    fpv.gy = count * rows; fpv.sy0 = 0u; fpv.sy1 = count * rows;
    fpv.cull_mode = 0u;
    c->front_fpv = fpv;
    const bool timing = (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0;
    const int sh_degree = c->cfg.sh_degree >= 0 ? c->cfg.sh_degree : sc->sh_degree_seen.load();
    const uint32_t tiles_v = c->gx * fpv.gy, stripe_tiles = sw * fpv.gy;
    const uint32_t nv = count * c->n_pad, blocks_v = count * fb.blocks;
    const int sig_bits = sig_bits_for(stripe_tiles ? stripe_tiles : 1u);
    KernelTimer *kt = c->kt.enabled ? &c->kt : nullptr;
    c->front_done = false;
    c->rounds_slot = -1;
    SceneSoA soa;
    int rc = wait_for_uploads(c, s, &soa);
    if (rc) return rc;
    // who evaluates the colours: the same rule as a plain frame, on the batch's totals (V and D_c scale together)
    bool lazy = c->last_lazy;
    if (sh_degree <= 0 || c->color_policy == 2) lazy = false;
    else if (c->color_policy == 1) lazy = true;
    else {
        const volatile uint32_t *h = c->hint_host;
        const uint32_t v_prev = h[0], dc_prev = h[1], posted = h[2];
        if (posted < 2u) lazy = true;
        else if ((uint64_t)dc_prev * 4u < (uint64_t)v_prev * 9u) lazy = true;
        else if ((uint64_t)dc_prev * 4u > (uint64_t)v_prev * 11u) lazy = false;
    }
    c->front_lazy = lazy;
    const bool geo = lazy && sh_degree > 0 && c->geo_policy == 1;
    c->front_geo = geo;
    if (geo && (rc = ensure_geo(c)) != GSPLAT_OK) return rc;
    if (sh_degree > 0 && soa.sh_block == nullptr) {
        std::lock_guard<std::mutex> lock(sc->mutex);
        if ((rc = ensure_slots(sc)) != GSPLAT_OK) return rc;
        HIP_TRY(hipStreamWaitEvent(s, sc->upload_done, 0));
        soa = sc->soa;
    }
    if (!lazy && (rc = ensure_culled(c)) != GSPLAT_OK) return rc;
    if (timing) HIP_TRY(hipEventRecord(c->ev[0], s));
    c->kt.begin(s);
    launch_project_batch(soa, c->n, fb, fpv, lazy ? (geo ? -2 : -1) : sh_degree, geo ? c->geo : c->culled, c->keys, c->block_sums,
                         c->sort.splat_hist, block_bounds, c->block_skip, c->tile_staged, tiles_v, c->counters->dc_parts,
                         scheduled_tiles(c, fpv), s, c->use_live_lists ? c->live_lists : nullptr, sort_splat_part_blocks(nv));
    if (kt) kt->mark(GSPLAT_KERNEL_PROJECT);
    if (timing) HIP_TRY(hipEventRecord(c->ev[1], s));
    const uint32_t *skip_marks = (block_bounds != nullptr && cull_mode != 0u) ? c->block_skip : nullptr;
    c->front_skip_marks = skip_marks != nullptr;
    launch_sort_splats(c->sort, c->keys, nv, skip_marks, s, kt, c->use_live_lists ? c->live_lists : nullptr);
    if (timing) HIP_TRY(hipEventRecord(c->ev[2], s));
    launch_emit_sums(c->sort.list[0], c->sort.v_count, nv, c->emit_sums, s);
    launch_scan_blocks(c->emit_sums, c->block_sums, blocks_v, c->block_base, c->capacity, &c->counters->total_emitted,
                       &c->counters->d_sorted, &c->counters->overflow, &c->counters->visible, c->counters->batch_last_tile,
                       c->bounds, (uint32_t)bounds_entries(c->gx, c->gy * c->batch), &c->counters->big_count, c->hint_dev,
                       c->counters->dc_parts, c->hint_dev ? c->hint_dev + 4 : nullptr, last_tiles_copy,
                       &c->counters->long_count, &c->counters->big_seen, s, fb.blocks);
    if (kt) kt->mark(GSPLAT_KERNEL_SCAN);
    const bool list_bigs = c->bigs_unknown > 0 ||
                           (c->hint_host != nullptr && reinterpret_cast<const volatile uint32_t *>(c->hint_host)[3] != 0u);
    if (c->bigs_unknown > 0) --c->bigs_unknown;
    c->front_big_hint = c->hint_host != nullptr ? reinterpret_cast<const volatile uint32_t *>(c->hint_host)[3] : 0u;
    launch_emit(c->sort.list[0], c->sort.v_count, nv, fpv, c->emit_sums, c->block_base, c->capacity, c->sort.keys[0],
                c->sort.values[0], &c->counters->big_count, c->big_list, /*narrow=*/true, s, 1, list_bigs, c->front_big_hint);
    c->front_list_bigs = list_bigs;
    if (kt) kt->mark(GSPLAT_KERNEL_EMIT);
    if (timing) HIP_TRY(hipEventRecord(c->ev[3], s));
    uint32_t wide_bins = 0;
    if (c->pair_sort_policy != 1) {
        wide_bins = sort_wide_bins(stripe_tiles);
        const uint32_t pairs_prev = c->hint_host ? reinterpret_cast<const volatile uint32_t *>(c->hint_host)[4] : 0u;
        if (c->pair_sort_policy == 0 && pairs_prev > WIDE_AUTO_PAIRS) wide_bins = 0;
    }
    if (!c->sort.rank_atomic || c->sort.wide_bins_allocated < wide_bins) wide_bins = 0;
    c->sorted_index = wide_bins ? launch_sort_pairs_wide(c->sort, &c->counters->d_sorted, c->capacity, wide_bins, s, kt)
                                : launch_sort_pairs(c->sort, &c->counters->d_sorted, c->capacity, sig_bits, s, kt, 16, true);
    c->front_wide_bins = wide_bins;
    c->front_narrow = true;
    c->front_rounds = false;
    c->front_stripe_cull = stripe_cull;
    c->last_frame = frames[count - 1u];
    if (timing) HIP_TRY(hipEventRecord(c->ev[4], s));
    HIP_TRY(hipGetLastError());
    c->front_fp = fb.f[count - 1u];
    c->front_soa = soa;
    c->front_sig_bits = sig_bits;
    c->front_sh_degree = sh_degree;
    c->front_done = true;
    c->rendered = false;
    return GSPLAT_OK;
}

static int batch_back(gsplat_ctx *c, const uint32_t *last_tiles_dev) {
    if (!c->front_done || c->front_batch.count == 0u) return GSPLAT_ERR_INVALID_ARGUMENT;
    hipStream_t s = c->stream;
    const FrameBatch &fb = c->front_batch;
    const FrameParams &fpv = c->front_fpv;
    const bool timing = (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0;
    KernelTimer *kt = c->kt.enabled ? &c->kt : nullptr;
    const bool fix_last = (c->cfg.flags & GSPLAT_FLAG_FIX_LAST_TILE) != 0;
    const uint32_t *last_tiles = last_tiles_dev ? last_tiles_dev : c->counters->batch_last_tile;
    const bool fast_exp = (c->cfg.flags & GSPLAT_FLAG_FAST_EXP) != 0;
    const int lazy_degree = c->front_lazy ? c->front_sh_degree : 0;
    const bool geo = c->front_geo && lazy_degree > 0;
    launch_boundaries_batch(c->sort.keys[c->sorted_index], &c->counters->d_sorted, c->bounds, fix_last, is_sharded(c), last_tiles,
                            fb, tile_map_of(fpv), s);
    c->values_index = c->sorted_index;
    if (kt) kt->mark(GSPLAT_KERNEL_BOUNDARIES);
    if (timing) HIP_TRY(hipEventRecord(c->ev[5], s));
    launch_render_batch(geo ? c->geo : c->culled, c->front_soa.sh_block, lazy_degree, c->sort.values[c->values_index], c->bounds,
                        fpv, fb, c->image, c->width, 0, 0, c->tile_staged, scheduled_tiles(c, fpv), fast_exp, geo, s);
    if (kt) kt->mark(GSPLAT_KERNEL_RENDER);
    if (timing) HIP_TRY(hipEventRecord(c->ev[6], s));
    HIP_TRY(hipGetLastError());
    c->timing_valid = timing;
    c->last_rounds = false;
    c->taps_stale = false;
    c->last_stripe_cull = c->front_stripe_cull;
    c->last_skip_marks = c->front_skip_marks;
    c->last_sig_bits = c->front_sig_bits;
    c->last_sh_degree = c->front_sh_degree;
    c->last_lazy = c->front_lazy;
    c->last_geo = c->front_geo;
    c->last_narrow = true;
    c->last_wide_bins = c->front_wide_bins;
    c->last_fp = c->front_fp;
    c->last_soa = c->front_soa;
    c->last_batch = fb.count;
    c->last_image = c->image;
    c->front_done = false;
    c->front_batch.count = 0u;
    c->rendered = true;
    return GSPLAT_OK;
}

extern "C++" {
namespace gsplat {
int ctx_batch_begin(gsplat_ctx *c, const gsplat_frame *frames, uint32_t count, uint32_t *last_tiles_out_device, bool stripe_cull) {
    if (!c || !frames) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    return batch_front(c, frames, count, stripe_cull, last_tiles_out_device);
}
uint32_t ctx_batch_capacity(const gsplat_ctx *c) { return c->batch; }
}  // namespace gsplat
}  // extern "C++"

int gsplat_create_batch_view(gsplat_ctx *owner, const gsplat_config *config, uint32_t batch, gsplat_ctx **out_ctx) {
    if (!owner || !config || !out_ctx) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    if (batch < 1u || batch > (uint32_t)MAX_BATCH) return GSPLAT_ERR_OUT_OF_RANGE;
    int rc = check_config(config);
    if (rc) return rc;
    if (config->max_splats != 0 && config->max_splats != owner->n) return GSPLAT_ERR_INVALID_ARGUMENT;
    const uint32_t factor = config->key_budget_factor ? config->key_budget_factor : 10u;
    if ((uint64_t)factor * owner->n * batch >= 0xFFFFF000ull) return GSPLAT_ERR_OUT_OF_RANGE;   // pair indices are 32-bit
    if ((uint64_t)owner->n_pad * batch >= 0xFFFFF000ull) return GSPLAT_ERR_OUT_OF_RANGE;       // ... and so are virtual slots
    if (batch > 1u && (config->flags & GSPLAT_FLAG_KEEP_EMITTED)) return GSPLAT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(owner->device));
    return ctx_create(config, owner->scene, owner->device, out_ctx, batch);
}

int gsplat_render_batch_begin(gsplat_ctx *c, const gsplat_frame *frames, uint32_t count, uint32_t *last_tiles_out_device) {
    return gsplat::ctx_batch_begin(c, frames, count, last_tiles_out_device, /*stripe_cull=*/true);
}

int gsplat_render_batch_end(gsplat_ctx *c, const uint32_t *frame_last_tiles_device) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    return batch_back(c, frame_last_tiles_device);
}

int gsplat_render_batch(gsplat_ctx *c, const gsplat_frame *frames, uint32_t count) {
    if (!c || !frames) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    // one call, no exchange: a stripe context may only skip what cannot change its frames' "last tile" words
    const int rc = batch_front(c, frames, count, /*stripe_cull=*/!is_sharded(c), nullptr);
    if (rc != GSPLAT_OK) return rc;
    return batch_back(c, nullptr);
}

int gsplat_batch_image_device_ptr(gsplat_ctx *c, uint32_t index, float **out_ptr) {
    if (!c || !out_ptr) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (index >= c->batch) return GSPLAT_ERR_OUT_OF_RANGE;
    *out_ptr = reinterpret_cast<float *>(c->image + (size_t)index * c->width * c->height);
    return GSPLAT_OK;
}

int gsplat_render(gsplat_ctx *c, const gsplat_frame *frame, float *rgba_out) {
    if (!c || !frame) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    float4 *target = default_target(c);
    bool copy_to_host = false;
    if (rgba_out) {
        if (is_device_pointer(rgba_out)) target = reinterpret_cast<float4 *>(rgba_out);
        else copy_to_host = true;
    }
    const int rc = render_impl(c, frame, target, c->width, 0, 0);
    if (rc != GSPLAT_OK) return rc;
    c->last_image = rgba_out && !copy_to_host ? nullptr : target;
    if (copy_to_host) {
        HIP_TRY(hipMemcpyAsync(rgba_out, target, (size_t)c->width * c->height * sizeof(float4),
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return GSPLAT_OK;
}

int gsplat_render_to(gsplat_ctx *c, const gsplat_frame *frame, float *device_out, uint32_t pitch_px, uint32_t origin_x,
                     uint32_t origin_y) {
    if (!c || !frame || !device_out || pitch_px == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (origin_x > c->sx0 * TILE || origin_y > c->sy0 * TILE) return GSPLAT_ERR_OUT_OF_RANGE;
    const uint32_t x_end = c->sx1 * TILE < c->width ? c->sx1 * TILE : c->width;
    if (x_end > origin_x && x_end - origin_x > pitch_px) return GSPLAT_ERR_OUT_OF_RANGE;
    HIP_TRY(hipSetDevice(c->device));
    c->last_image = nullptr;  // (caller-owned memory: the image tap has nothing of this frame)
    return render_impl(c, frame, reinterpret_cast<float4 *>(device_out), pitch_px, origin_x, origin_y);
}

int gsplat_render_begin(gsplat_ctx *c, const gsplat_frame *frame, uint32_t *last_tile_out_device) {
    return gsplat::ctx_render_begin(c, frame, last_tile_out_device, /*stripe_cull=*/true);
}

int gsplat_render_end(gsplat_ctx *c, float *device_out, uint32_t pitch_px, uint32_t origin_x, uint32_t origin_y,
                      const uint32_t *frame_last_tile_device) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    if (!device_out) {
        c->last_image = default_target(c);
        return render_back(c, c->last_image, c->width, 0, 0, frame_last_tile_device);
    }
    if (pitch_px == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (origin_x > c->sx0 * TILE || origin_y > c->sy0 * TILE) return GSPLAT_ERR_OUT_OF_RANGE;
    const uint32_t x_end = c->sx1 * TILE < c->width ? c->sx1 * TILE : c->width;
    if (x_end > origin_x && x_end - origin_x > pitch_px) return GSPLAT_ERR_OUT_OF_RANGE;
    return render_back(c, reinterpret_cast<float4 *>(device_out), pitch_px, origin_x, origin_y,
                       frame_last_tile_device);
}

int gsplat_pick(gsplat_ctx *c, const gsplat_frame *frame, uint32_t tile_id, float out_xyzn[4]) {
    if (!c || !frame || !out_xyzn) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (!c->rendered) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (c->last_batch != 0u) return GSPLAT_ERR_UNSUPPORTED;  // (the last launch sequence composited a batch: render the frame alone)
    if (tile_id >= c->gx * c->gy) return GSPLAT_ERR_OUT_OF_RANGE;
    HIP_TRY(hipSetDevice(c->device));
    {   // the pick walks the tile's complete sorted list: a two-round frame is replayed in one round first
        const int rc = replay_full(c);
        if (rc != GSPLAT_OK) return rc;
    }
    hipStream_t s = c->stream;
    // get_splat_position re-runs gsplat_render.glsl alone (gaussian_splatting_rasterizer.gd:166-168): the compositor sees
    // the LAST frame's buffers and uniforms and only its two push constants are new.  A lazy frame's compositor rebuilds
    // the records it stages from FrameParams, so those must be the rendered frame's own — a pick with a later clock,
    // camera or model scale would blend new geometry against the old tile lists.
    FrameParams fp = c->last_fp;
    fp.heatmap_factor = frame->heatmap_factor;
    fp.target_tile = tile_id;
    // gaussian_splatting_rasterizer.gd:166-168 re-runs the whole compositor; only the target tile can write
    // the pick record, so a 1x1 grid on that tile gives the same 16 bytes.
    const uint32_t tx = tile_id % c->gx, ty = tile_id / c->gx;
    if (tx < c->sx0 || tx >= c->sx1 || ty < c->sy0 || ty >= c->sy1) return GSPLAT_ERR_OUT_OF_RANGE;
    fp.sx0 = tx; fp.sx1 = tx + 1; fp.sy0 = ty; fp.sy1 = ty + 1;
    // the pick re-composites the target tile INTO the last frame's image; if that is an image of the asynchronous ring,
    // its copy to the host may still be in flight on the ring's stream — the tile must not change under it (a pick with
    // another heat-map factor would tear the host image of the previous ticket)
    if (c->async.ready && c->async.count > 0 && c->last_image != nullptr &&
        (c->last_image == c->async.dev[0] || c->last_image == c->async.dev[1]))
        HIP_TRY(hipStreamWaitEvent(s, c->async.copy_done[(c->async.count - 1) % 3u], 0));
    HIP_TRY(hipMemsetAsync(c->pick, 0, sizeof(float4), s));  // no stale hits
    const bool pick_geo = c->last_lazy && c->last_geo && c->last_sh_degree > 0;
    launch_render(pick_geo ? c->geo : c->culled, c->last_soa.sh_block, c->last_lazy ? c->last_sh_degree : 0,
                  c->sort.values[c->values_index], c->bounds, fp, c->last_image ? c->last_image : c->image, c->width, 0, 0,
                  c->pick, nullptr, TileSchedule{},
                  (c->cfg.flags & GSPLAT_FLAG_FAST_EXP) != 0, s, 0, nullptr, nullptr, nullptr, pick_geo);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_xyzn, c->pick, sizeof(float4), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GSPLAT_OK;
}

int gsplat_get_stats(gsplat_ctx *c, gsplat_stats *user_out) {
    if (!c || !user_out) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (user_out->struct_size < 2 * sizeof(uint32_t)) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof h, hipMemcpyDeviceToHost));
    // the caller says how large ITS gsplat_stats is: never write past that (a binding built against an older header)
    const uint32_t caller_size = user_out->struct_size;
    gsplat_stats full;
    gsplat_stats *out = &full;
    memset(out, 0, sizeof *out);
    out->struct_size = caller_size < sizeof(gsplat_stats) ? caller_size : (uint32_t)sizeof(gsplat_stats);
    out->num_splats = c->n;
    out->num_visible = h.visible;
    out->num_emitted = h.total_emitted;
    out->num_sorted = h.total_emitted < c->capacity ? h.total_emitted : c->capacity;  // min(D, capacity), main.gd:97-100
    if (c->last_rounds) {
        out->pairs_round[0] = h.round_total[0] < c->capacity ? h.round_total[0] : c->capacity;
        out->pairs_round[1] = h.plan.single ? 0 : h.round_total[1];
    }
    else { out->pairs_round[0] = out->num_sorted; out->pairs_round[1] = 0; }
    {   // D_c = sum over this context's tiles of the pairs the compositor staged
        const size_t tiles = (size_t)c->gx * c->gy;
        std::vector<uint32_t> staged(tiles);
        HIP_TRY(hipMemcpy(staged.data(), c->tile_staged, tiles * 4, hipMemcpyDeviceToHost));
        uint64_t dc = 0;
        const uint32_t ty0 = c->last_batch ? 0u : c->sy0, ty1 = c->last_batch ? c->last_batch * (c->sy1 - c->sy0) : c->sy1;
        if (c->last_batch) staged.resize((size_t)c->gx * ty1);  // (a batch: the virtual grid's rows)
        if (c->last_batch) HIP_TRY(hipMemcpy(staged.data(), c->tile_staged, staged.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t ty = ty0; ty < ty1; ++ty)
            for (uint32_t tx = c->sx0; tx < c->sx1; ++tx) dc += staged[(size_t)ty * c->gx + tx];
        out->num_composited = c->rendered ? dc : 0;
    }
    out->capacity = c->capacity;
    out->overflow = (int32_t)h.overflow;
    // two on the splats' depth16 + the tile bits of the pairs (one pass in the counting-sort form)
    out->sort_passes = 2 + (c->last_wide_bins ? 1 : (c->last_sig_bits > 16 ? sort_num_passes(c->last_sig_bits - 16) : 0));
    out->sh_degree = c->last_sh_degree;
    out->lazy_colors = c->last_lazy ? 1 : 0;
    out->pair_key_bytes = c->last_narrow ? 2 : 4;
    out->scene_bytes = c->scene->bytes;
    out->bytes_allocated = c->bytes_allocated + c->scene->bytes;
    if (c->timing_valid) {
        float a = 0.0f, b = 0.0f;
        HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, c->ev[2], c->ev[3]));
        out->ms_projection = a + b;  // projection kernel + key emission (the reference's 'Projection' pass)
        HIP_TRY(hipEventElapsedTime(&a, c->ev[1], c->ev[2]));
        HIP_TRY(hipEventElapsedTime(&b, c->ev[3], c->ev[4]));
        out->ms_sort = a + b;        // splat-level passes + pair-level passes
        HIP_TRY(hipEventElapsedTime(&out->ms_boundaries, c->ev[4], c->ev[5]));
        HIP_TRY(hipEventElapsedTime(&out->ms_render, c->ev[5], c->ev[6]));
        HIP_TRY(hipEventElapsedTime(&out->ms_total, c->ev[0], c->ev[6]));
    }
    if (c->kt.enabled && c->rendered) {
        for (int i = 0; i < c->kt.count; ++i) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, c->kt.ev[i], c->kt.ev[i + 1]));
            out->ms_kernel[c->kt.cls[i]] += ms;
            out->launches_kernel[c->kt.cls[i]] += 1;
        }
    }
    // SURVEY.md §8(d) algorithmic bytes; K = coefficients per channel evaluated.  The 12 K bytes of SH coefficients are
    // counted for the colours this build evaluated: every visible splat in an eager frame, every staged pair in a lazy one.
    const uint64_t N = c->n, V = h.visible, D = out->num_sorted;
    const uint64_t K = (uint64_t)(c->last_sh_degree + 1) * (c->last_sh_degree + 1);
    const uint64_t T = (uint64_t)c->gx * c->gy, P = (uint64_t)c->width * c->height;
    const uint64_t evaluated = c->last_lazy ? out->num_composited : V;
    out->algorithmic_bytes[0] = 16 * N + 28 * V + 12 * K * evaluated + 48 * V + 8 * D;
    out->algorithmic_bytes[1] = 4 * D + 4 * 16 * D;  // 68 D: the reference's four pair passes (this build moves less)
    out->algorithmic_bytes[2] = 4 * D + 8 * T;
    out->algorithmic_bytes[3] = 40 * D + 16 * P;
    if (c->timing_valid && c->async.ready && c->async.last_waited) {
        const int hs = (int)((c->async.last_waited - 1) % 3u);
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->async.copy_start[hs], c->async.copy_done[hs]) == hipSuccess) out->ms_readback = ms;
        else (void)hipGetLastError();
    }
    if (c->timing_valid && c->gather_start && c->gather_stop) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->gather_start, c->gather_stop) == hipSuccess) out->ms_gather = ms;
        else (void)hipGetLastError();
    }
    memcpy(user_out, out, out->struct_size);
    return GSPLAT_OK;
}

int gsplat_set_timing(gsplat_ctx *c, uint32_t timing_flags) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t mask = GSPLAT_FLAG_TIMING | GSPLAT_FLAG_KERNEL_TIMING;
    c->cfg.flags = (c->cfg.flags & ~mask) | (timing_flags & mask);
    const bool want_kt = (c->cfg.flags & GSPLAT_FLAG_KERNEL_TIMING) != 0;
    if (want_kt && !c->kt_events_created) {
        for (int i = 0; i <= KernelTimer::MAX_MARKS; ++i) HIP_TRY(hipEventCreate(&c->kt.ev[i]));
        c->kt_events_created = true;
    }
    c->kt.enabled = want_kt;
    c->kt.count = 0;
    c->timing_valid = false;
    return GSPLAT_OK;
}

int gsplat_debug_read(gsplat_ctx *c, int which, void *dst, size_t size, size_t *bytes_written) {
    if (!c || (!dst && size)) return GSPLAT_ERR_INVALID_ARGUMENT;
    SceneStore *sc = c->scene.get();
    HIP_TRY(hipSetDevice(c->device));
    if (c->last_batch != 0u && which != GSPLAT_DEBUG_IMAGE && which != GSPLAT_DEBUG_RECORDS && which != GSPLAT_DEBUG_SLOT_IDS &&
        which != GSPLAT_DEBUG_SORT_RANK && which != GSPLAT_DEBUG_EMIT_MODE)
        return GSPLAT_ERR_UNSUPPORTED;  // (the intermediate arrays hold a BATCH's virtual frame: render a frame alone for its taps)
    if (which == GSPLAT_DEBUG_KEYS_SORTED || which == GSPLAT_DEBUG_VALUES_SORTED || which == GSPLAT_DEBUG_TILE_BOUNDS) {
        const int rc = replay_full(c);  // a two-round frame holds round B's arrays only
        if (rc != GSPLAT_OK) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof h, hipMemcpyDeviceToHost));
    const void *src = nullptr;
    size_t avail = 0;
    // temporaries of the taps: freed on every way out of this function, the HIP_TRY returns included
    struct Scratch {
        float *p = nullptr;
        ~Scratch() { if (p) (void)hipFree(p); }
    } scratch, scratch2;
    float *&tmp = scratch.p, *&tmp2 = scratch2.p;
    // a re-laid-out scene keeps per-splat arrays in storage order and slot numbers in the value arrays: the taps
    // present everything in splat-id terms, like a context on a scene that was never finalized
    auto mapped_u32 = [&](const uint32_t *srcp, const uint32_t *index, size_t count) -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (count ? count : 4) * 4));
        launch_gather_u32(srcp, reinterpret_cast<uint32_t *>(tmp), index, (uint32_t)count, c->stream);
        HIP_TRY(hipStreamSynchronize(c->stream));
        src = tmp;
        avail = count * 4;
        return GSPLAT_OK;
    };
    switch (which) {
        case GSPLAT_DEBUG_CULLED: {
            const int erc = ensure_culled(c);
            if (erc != GSPLAT_OK) return erc;
        }
            avail = (size_t)c->n * 48;
            // a lazy frame writes no records (its compositor recomputes what it stages from the scene); the tap shows the
            // reference's full record of every visible splat
            if (c->rendered && c->last_lazy) {
                launch_fill_records(c->last_soa, c->n, c->last_fp, c->last_sh_degree, c->culled, c->stream);
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            if (sc->finalized) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), avail ? avail : 16));
                launch_gather_raster(c->culled, reinterpret_cast<float4 *>(tmp), sc->slot_of_id, c->n, c->stream);
                HIP_TRY(hipStreamSynchronize(c->stream));
                src = tmp;
            } else {
                src = c->culled;
            }
            break;
        case GSPLAT_DEBUG_KEYS_SORTED:
            avail = (size_t)h.d_sorted * 4;
            if (c->last_narrow) {  // the frame sorted 16-bit tile ids: present the reference's keys
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), avail ? avail : 16));
                launch_widen_keys(reinterpret_cast<const uint16_t *>(c->sort.keys[c->sorted_index]),
                                  c->sort.values[c->values_index], c->keys.key, &c->counters->d_sorted,
                                  reinterpret_cast<uint32_t *>(tmp), tile_map_of(c->last_fp), c->stream);
                HIP_TRY(hipStreamSynchronize(c->stream));
                src = tmp;
            } else {
                src = c->sort.keys[c->sorted_index];
            }
            break;
        case GSPLAT_DEBUG_VALUES_SORTED:
            if (sc->finalized) {  // value v is a slot: present id_of_slot[v]
                const int rc = mapped_u32(sc->id_of_slot, c->sort.values[c->values_index], h.d_sorted);
                if (rc) return rc;
            } else {
                src = c->sort.values[c->values_index];
                avail = (size_t)h.d_sorted * 4;
            }
            break;
        case GSPLAT_DEBUG_TILE_BOUNDS: src = c->bounds; avail = (size_t)c->gx * c->gy * 8; break;
        case GSPLAT_DEBUG_KEYS_EMITTED:
        case GSPLAT_DEBUG_VALUES_EMITTED:
            // ping-pong half 0 is overwritten by the second pair pass: needs GSPLAT_FLAG_KEEP_EMITTED.  After
            // gsplat_finalize_scene equal depth codes are emitted in storage order, not ascending splat id.
            if (!c->emit_keys) return GSPLAT_ERR_UNSUPPORTED;
            if (which == GSPLAT_DEBUG_VALUES_EMITTED && sc->finalized) {
                const int rc = mapped_u32(sc->id_of_slot, c->emit_values, h.d_sorted);
                if (rc) return rc;
            } else {
                src = which == GSPLAT_DEBUG_KEYS_EMITTED ? c->emit_keys : c->emit_values;
                avail = (size_t)h.d_sorted * 4;
            }
            break;
        case GSPLAT_DEBUG_TILE_COUNTS: {  // num_tiles_touched = w * h of the projection hand-off
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp2), ((size_t)c->n ? (size_t)c->n : 4) * 4));
            launch_tile_counts(c->keys.dims, reinterpret_cast<uint32_t *>(tmp2), c->n,
                               (c->front_done ? c->front_skip_marks : c->last_skip_marks) ? c->block_skip : nullptr, c->stream);
            if (sc->finalized) {
                const int rc = mapped_u32(reinterpret_cast<uint32_t *>(tmp2), sc->slot_of_id, c->n);
                if (rc) return rc;
            } else {
                HIP_TRY(hipStreamSynchronize(c->stream));
                src = tmp2;
                avail = (size_t)c->n * 4;
            }
            break;
        }
        case GSPLAT_DEBUG_TILE_STAGED: src = c->tile_staged; avail = (size_t)c->gx * c->gy * 4; break;
        case GSPLAT_DEBUG_TILE_ORDER:
            src = c->tile_order; avail = (size_t)scheduled_tiles(c, c->last_fp).entries * 4;
            break;
        case GSPLAT_DEBUG_BLOCK_SUMS: src = c->block_sums; avail = (size_t)sc->num_proj_blocks * 16; break;
        case GSPLAT_DEBUG_SORT_RANK: {
            const uint32_t mode = c->sort.rank_atomic ? 1u : 0u;
            if (bytes_written) *bytes_written = sizeof(mode);
            if (size < sizeof(mode)) return GSPLAT_ERR_INVALID_ARGUMENT;
            memcpy(dst, &mode, sizeof(mode));
            return GSPLAT_OK;
        }
        case GSPLAT_DEBUG_EMIT_MODE: {
            const uint32_t mode = c->front_list_bigs ? 1u : 0u;
            if (bytes_written) *bytes_written = sizeof(mode);
            if (size < sizeof(mode)) return GSPLAT_ERR_INVALID_ARGUMENT;
            memcpy(dst, &mode, sizeof(mode));
            return GSPLAT_OK;
        }
        case GSPLAT_DEBUG_IMAGE:
            src = c->last_image ? c->last_image : c->image;
            avail = (size_t)c->width * c->height * 16 * (c->last_batch ? c->last_batch : 1u);  // (a batch: its images, one after the other)
            break;
        case GSPLAT_DEBUG_SLOT_IDS: {
            if (sc->finalized) {
                src = sc->id_of_slot;
                avail = (size_t)c->n * 4;
            } else {  // upload order: slot s holds splat s
                std::vector<uint32_t> ident(c->n);
                for (uint32_t i = 0; i < c->n; ++i) ident[i] = i;
                const size_t nb = std::min(size, (size_t)c->n * 4);
                if (nb) memcpy(dst, ident.data(), nb);
                if (bytes_written) *bytes_written = nb;
                return GSPLAT_OK;
            }
            break;
        }
        case GSPLAT_DEBUG_RECORDS: {
            HIP_TRY(hipStreamSynchronize(sc->upload_stream));
            avail = (size_t)c->n * 240;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), avail ? avail : 16));
            launch_gather_records(sc->soa, c->n, tmp, sc->finalized ? sc->slot_of_id : nullptr, c->stream);
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) return hip_fail(e, "gather", __FILE__, __LINE__);
            src = tmp;
            break;
        }
        default: return GSPLAT_ERR_INVALID_ARGUMENT;
    }
    const size_t nbytes = size < avail ? size : avail;
    int rc = GSPLAT_OK;
    if (nbytes) {
        hipError_t e = hipMemcpy(dst, src, nbytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
    }
    if (bytes_written) *bytes_written = nbytes;
    return rc;
}

int gsplat_debug_pow02(gsplat_ctx *c, uint32_t first_bits, uint64_t count, float *out_host) {
    if (!c || (!out_host && count)) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (!count) return GSPLAT_OK;
    HIP_TRY(hipSetDevice(c->device));
    float *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)count * sizeof(float)));
    launch_pow02_bits(first_bits, count, tmp, c->stream);
    hipError_t e = hipMemcpyAsync(out_host, tmp, (size_t)count * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "gsplat_debug_pow02", __FILE__, __LINE__);
    return GSPLAT_OK;
}

int gsplat_image_device_ptr(gsplat_ctx *c, float **out_ptr) {
    if (!c || !out_ptr) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out_ptr = reinterpret_cast<float *>(default_target(c));
    return GSPLAT_OK;
}

// ---- hand-off of finished frames -----------------------------------------------------------------------------
int gsplat_render_async(gsplat_ctx *c, const gsplat_frame *frame, uint64_t *ticket_out) {
    if (!c || !frame || !ticket_out) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    gsplat_ctx::AsyncRing &a = c->async;
    // GSPLAT_FLAG_READBACK_RGB: 12 bytes per pixel cross PCIe instead of 16 — alpha is the constant 1.0 of
    // gsplat_render.glsl:101 — so a 1080p frame is 24.9 MB / 0.45 ms instead of 33.2 MB / 0.60 ms and the pipelined rate is
    // bound by the frame again, not by the link (Godot side: Image.FORMAT_RGBF)
    const bool rgb = (c->cfg.flags & GSPLAT_FLAG_READBACK_RGB) != 0;
    const size_t bytes = (size_t)c->width * c->height * (rgb ? 3 : 4) * sizeof(float);
    if (!a.ready) {
        // two device images of the ring's own (round 3 let dev[0] alias the context's image: a synchronous frame or a pick
        // between two asynchronous ones then overwrote an image whose copy to the host was still in flight)
        auto setup = [&]() -> int {
            int rc;
            for (float4 *&dimg : a.dev)
                if ((rc = dev_alloc(c, &dimg, (size_t)c->width * c->height, true)) != GSPLAT_OK) return rc;
            if (rgb)
                for (float *&drgb : a.dev_rgb)
                    if ((rc = dev_alloc(c, &drgb, (size_t)c->width * c->height * 3, true)) != GSPLAT_OK) return rc;
            HIP_TRY(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
            for (float *&h : a.host) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h), bytes, hipHostMallocDefault));
            for (hipEvent_t &e : a.rendered) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (hipEvent_t &e : a.copy_start) HIP_TRY(hipEventCreate(&e));
            for (hipEvent_t &e : a.copy_done) HIP_TRY(hipEventCreate(&e));
            return GSPLAT_OK;
        };
        const int rc = setup();
        if (rc != GSPLAT_OK) {
            release_async(c);  // nothing half-made stays behind: the next call starts over
            return rc;
        }
        a.ready = true;
    }
    const uint64_t n = a.count;
    const int d = (int)(n & 1u), hs = (int)(n % 3u);
    // device image d was the source of the copy of frame n - 2: the compositor must not overwrite it before that copy is
    // through (a stream-side wait; by then it normally is)
    if (n >= 2) HIP_TRY(hipStreamWaitEvent(c->stream, a.copy_done[(n - 2) % 3u], 0));
    const int rc = render_impl(c, frame, a.dev[d], c->width, 0, 0);
    if (rc != GSPLAT_OK) return rc;
    c->last_image = a.dev[d];
    if (rgb) {  // (behind the compositor on the context's stream; dev_rgb[d] is free: the copy of frame n - 2 was waited for above)
        launch_pack_rgb(a.dev[d], c->width, c->width, c->height, a.dev_rgb[d], c->stream);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(a.rendered[d], c->stream));
    HIP_TRY(hipStreamWaitEvent(a.stream, a.rendered[d], 0));
    HIP_TRY(hipEventRecord(a.copy_start[hs], a.stream));
    // (the runtime's copy engine: measured against a copy kernel of the library's own — a few workgroups streaming the image
    // to mapped pinned memory — which HALVED the pipelined rate: stores of the CUs to PCIe back the chip's write path up and
    // every kernel that overlaps the copy runs at the link's pace; experiments/readback_kernel.patch, profiles/r05_d2h_probe_*)
#ifdef GSPLAT_TEST_HOOKS
    // (diagnosis builds only — `build.py variant <name> api.hip -DGSPLAT_TEST_HOOKS`: the ring's events without its copy,
    // experiments/README.md.  The host image is then STALE, so the shipped library has no such switch; tests/test_host.py
    // greps for probe switches outside this guard)
    static const bool probe_no_copy = getenv("GSPLAT_PROBE_RING_NO_COPY") != nullptr;
#else
    constexpr bool probe_no_copy = false;
#endif
    if (!probe_no_copy)
        HIP_TRY(hipMemcpyAsync(a.host[hs], rgb ? static_cast<const void *>(a.dev_rgb[d]) : static_cast<const void *>(a.dev[d]),
                               bytes, hipMemcpyDeviceToHost, a.stream));
    HIP_TRY(hipEventRecord(a.copy_done[hs], a.stream));
    a.count = n + 1;
    *ticket_out = n + 1;
    return GSPLAT_OK;
}

int gsplat_readback_wait(gsplat_ctx *c, uint64_t ticket, const float **host_rgba_out) {
    if (!c || !host_rgba_out || ticket == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    gsplat_ctx::AsyncRing &a = c->async;
    if (!a.ready || ticket > a.count) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (ticket + 3 <= a.count) return GSPLAT_ERR_OUT_OF_RANGE;  // its host image has been handed to a later frame
    HIP_TRY(hipSetDevice(c->device));
    const int hs = (int)((ticket - 1) % 3u);
    HIP_TRY(hipEventSynchronize(a.copy_done[hs]));
    a.last_waited = ticket;
    *host_rgba_out = a.host[hs];
    return GSPLAT_OK;
}

int gsplat_bind_external_image(gsplat_ctx *c, int fd, uint64_t size_bytes, uint64_t offset_bytes) {
    // (gsplat.h: the library takes ownership of fd — on every way out, so a failed bind does not leak the descriptor)
    struct FdGuard {
        int fd;
        ~FdGuard() { if (fd >= 0) (void)close(fd); }
    } guard{fd};
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unbind_external(c);
    forget_history(c);
    if (fd < 0) return GSPLAT_OK;
    const uint64_t need = (uint64_t)c->width * c->height * sizeof(float4);
    if (offset_bytes % sizeof(float4) != 0 || offset_bytes > size_bytes || size_bytes - offset_bytes < need)
        return GSPLAT_ERR_OUT_OF_RANGE;
    hipExternalMemoryHandleDesc hd;
    memset(&hd, 0, sizeof hd);
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = fd;
    hd.size = size_bytes;
    hipExternalMemory_t mem = nullptr;
    HIP_TRY(hipImportExternalMemory(&mem, &hd));
    guard.fd = -1;  // imported: the runtime owns the descriptor now
    hipExternalMemoryBufferDesc bd;
    memset(&bd, 0, sizeof bd);
    bd.offset = offset_bytes;
    bd.size = need;
    void *ptr = nullptr;
    hipError_t e = hipExternalMemoryGetMappedBuffer(&ptr, mem, &bd);
    if (e != hipSuccess) {
        (void)hipDestroyExternalMemory(mem);
        return hip_fail(e, "hipExternalMemoryGetMappedBuffer", __FILE__, __LINE__);
    }
    c->ext_mem = mem;
    c->ext_image = static_cast<float4 *>(ptr);
    return GSPLAT_OK;
}

int gsplat_export_image_fd(gsplat_ctx *c, int *fd_out, uint64_t *size_bytes_out) {
    if (!c || !fd_out) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)c->width * c->height * sizeof(float4);
    int fd = -1;
    // (dma-buf export works on whole pages: the allocation behind the image is page-granular)
    size_t sizes[2] = {bytes, (bytes + 4095u) & ~(size_t)4095u};
    hipError_t e = hipErrorUnknown;
    size_t used = 0;
    for (size_t sz : sizes) {
        e = hipMemGetHandleForAddressRange(&fd, reinterpret_cast<hipDeviceptr_t>(c->image), sz,
                                           hipMemRangeHandleTypeDmaBufFd, 0);
        if (e == hipSuccess) { used = sz; break; }
        (void)hipGetLastError();
    }
    if (e != hipSuccess) {
        (void)hip_fail(e, "hipMemGetHandleForAddressRange", __FILE__, __LINE__);
        return GSPLAT_ERR_UNSUPPORTED;
    }
    *fd_out = fd;
    if (size_bytes_out) *size_bytes_out = used;
    return GSPLAT_OK;
}

int gsplat_synchronize(gsplat_ctx *c) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSPLAT_OK;
}

int gsplat_make_view_proj(const float cam[12], const float basis_override[9], float fovy_degrees, float aspect,
                          float z_near, float z_far, float out32[32], float out_cam_pos[3]) {
    if (!cam || !out32) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (!(aspect > 0.0f) || !(z_far > z_near) || !(fovy_degrees > 0.0f)) return GSPLAT_ERR_INVALID_ARGUMENT;
    // view := Transform3D(basis_override, 0) * camera transform (gaussian_splatting_rasterizer.gd:176)
    float X[3], Y[3], Z[3], O[3];
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const float *B = basis_override ? basis_override : I;  // columns
    const float *cols[4] = {cam, cam + 3, cam + 6, cam + 9};
    float *outs[4] = {X, Y, Z, O};
    for (int k = 0; k < 4; ++k)
        for (int r = 0; r < 3; ++r)
            outs[k][r] = (B[0 * 3 + r] * cols[k][0] + B[1 * 3 + r] * cols[k][1]) + B[2 * 3 + r] * cols[k][2];
    float *v = out32, *p = out32 + 16;
    // gaussian_splatting_rasterizer.gd:185-188
    v[0] = -X[0]; v[1] = Y[0]; v[2] = -Z[0]; v[3] = 0.0f;
    v[4] = -X[1]; v[5] = Y[1]; v[6] = -Z[1]; v[7] = 0.0f;
    v[8] = X[2]; v[9] = -Y[2]; v[10] = Z[2]; v[11] = 0.0f;
    v[12] = -((O[0] * X[0] + O[1] * X[1]) + O[2] * X[2]);
    v[13] = -((O[0] * -Y[0] + O[1] * -Y[1]) + O[2] * -Y[2]);
    v[14] = -((O[0] * Z[0] + O[1] * Z[1]) + O[2] * Z[2]);
    v[15] = 1.0f;
    // Godot 4.3 Projection::set_perspective (core/math/projection.cpp, third-party, not in the reference tree)
    const float radians = (fovy_degrees / 2.0f) * 0.017453292519943295f;
    const float sine = sinf(radians), delta_z = z_far - z_near;
    if (delta_z == 0.0f || sine == 0.0f) return GSPLAT_ERR_INVALID_ARGUMENT;
    const float cotangent = cosf(radians) / sine;
    memset(p, 0, 16 * sizeof(float));
    p[0] = cotangent / aspect;
    p[5] = cotangent;
    p[10] = -(z_far + z_near) / delta_z;
    p[11] = -1.0f;  // gaussian_splatting_rasterizer.gd:192 forces [2][3] = -1
    p[14] = -2.0f * z_near * z_far / delta_z;
    p[15] = 0.0f;   // :193 forces [3][3] = 0
    if (out_cam_pos) {  // gaussian_splatting_rasterizer.gd:125-126: basis_override * origin, then (-x,-y,z)
        out_cam_pos[0] = -O[0]; out_cam_pos[1] = -O[1]; out_cam_pos[2] = O[2];
    }
    return GSPLAT_OK;
}

const char *gsplat_status_string(int status) {
    switch (status) {
        case GSPLAT_OK: return "ok";
        case GSPLAT_ERR_INVALID_ARGUMENT: return "invalid argument";
        case GSPLAT_ERR_OUT_OF_MEMORY: return "out of device memory";
        case GSPLAT_ERR_HIP: return "HIP runtime error";
        case GSPLAT_ERR_NO_DEVICE: return "no HIP device";
        case GSPLAT_ERR_OUT_OF_RANGE: return "out of range";
        case GSPLAT_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

const char *gsplat_last_error(void) { return g_last_error; }

uint32_t gsplat_version(void) { return ((uint32_t)GSPLAT_VERSION_MAJOR << 16) | GSPLAT_VERSION_MINOR; }

}  // extern "C"
