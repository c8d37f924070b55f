// C ABI of libgsplat_hip.so (include/gsplat.h): context, device memory, frame orchestration.
// Host-side counterpart of util/gaussian_splatting_rasterizer.gd (init_gpu / rasterize /
// get_splat_position / texture_size setter / update_camera_matrices) with HIP streams and device-side
// counters instead of Vulkan descriptor sets, indirect dispatches and per-dispatch barriers.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/gsplat.h"
#include "gsplat_internal.h"

using namespace gsplat;

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

// device counters, one 64-byte block
struct Counters {
    uint64_t total_emitted;  // D before the clamp
    uint64_t composited;     // D_c
    uint32_t d_sorted;       // min(D, capacity): the pair count every later pass reads
    uint32_t overflow;
    uint32_t visible;
    uint32_t frame_last_tile_plus1;  // highest tile touched by any splat's unclamped rectangle, +1
    uint32_t sort_error;     // look-back spin bound hit (must stay 0)
    uint32_t sh_degree_max;  // running max over uploads (not cleared per frame)
    uint32_t tickets[4];     // onesweep partition tickets
    uint32_t proj_ticket;    // fused projection chunk tickets
    uint32_t big_count;      // splats listed for emit_big_kernel this frame
    uint32_t tile_big_count; // tiles listed for tile_sort_big_kernel this frame (zeroed together with big_count)
    uint32_t hint_frames;    // frames whose {V, D_c} the scan kernel has posted to the host (never cleared)
    uint32_t pad[14];
};

}  // namespace

struct gsplat_ctx {
    gsplat_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t upload_stream = nullptr;
    std::mutex upload_mutex;

    uint32_t n = 0;
    uint64_t capacity = 0;
    uint32_t width = 0, height = 0, gx = 0, gy = 0;
    uint32_t sx0 = 0, sx1 = 0, sy0 = 0, sy1 = 0;  // stripe in tiles
    int sh_degree_seen = 0;

    SceneSoA scene{};
    float4 *culled = nullptr;
    uint32_t *local_off = nullptr, *counts = nullptr, *depths = nullptr, *tile_staged = nullptr;
    uint4 *block_sums = nullptr;
    uint2 *rects = nullptr;
    uint64_t *block_base = nullptr;
    uint32_t *big_list = nullptr;  // splats covering > 512 tiles, written by emit_big_kernel
    bool fused_projection = false;
    unsigned long long *chunk_status = nullptr;
    uint2 *chunk_info = nullptr;
    SortBuffers sort{};
    uint32_t *emit_keys = nullptr, *emit_values = nullptr;  // GSPLAT_FLAG_KEEP_EMITTED
    uint2 *bounds = nullptr;
    float4 *image = nullptr;
    float4 *pick = nullptr;
    Counters *counters = nullptr;
    uint32_t num_proj_blocks = 0;
    uint64_t bytes_allocated = 0;

    int sorted_index = 0;  // which ping-pong half holds the sorted pairs (keys) of the last frame
    int values_index = 0;  // ... and the sorted values (differs from sorted_index after the tie fix-up)
    // scene re-layout (gsplat_finalize_scene): storage slot <-> splat id
    bool finalized = false;
    uint32_t *id_of_slot = nullptr, *slot_of_id = nullptr;
    float4 *block_bounds = nullptr;          // 3 float4 per projection workgroup (GSPLAT_FLAG_BLOCK_CULL)
    uint32_t *block_skip = nullptr;          // per frame: 1 = the workgroup cannot emit anything
    std::atomic<bool> bounds_dirty{false};   // an upload changed the stored scene after the bounds were taken
    FrameParams front_fp;                    // parameters of the frame gsplat_render_begin started
    FrameParams last_fp;                     // parameters of the last finished frame (parity taps)
    // where the SH colours are evaluated this frame: by the compositor for the splats it stages (lazy) or by the
    // projection pass for every visible splat (eager).  Chosen per frame from what the previous frames did.
    int color_policy = 0;                    // 0 auto, 1 always lazy, 2 always eager (GSPLAT_COLOR)
    bool front_lazy = false, last_lazy = false;
    uint32_t *hint_host = nullptr;           // host-mapped: {visible splats, pairs staged by the previous frame, frames}
    uint32_t *hint_dev = nullptr;            // the same three words as the device sees them
    uint2 *segs = nullptr;                   // tile-major sort: the tiles' true segments (lives behind `bounds`)
    uint32_t *tile_big_list = nullptr;       // tiles with more than 4096 pairs (tile_sort_big_kernel)
    bool tile_timing_valid = false;
    bool tile_major_sort = false;            // GSPLAT_SORT=tile: two global passes on the tile bits + per-tile depth sort
    hipEvent_t ev_tile[2] = {nullptr, nullptr};  // around the per-tile depth sort (its time counts as sort time)
    bool front_done = false;
    int front_sig_bits = 0, front_sh_degree = 0;
    int last_sig_bits = 32;
    int last_sh_degree = 0;
    bool rendered = false;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timing_valid = false;
    KernelTimer kt;
    bool kt_events_created = false;

    std::vector<void *> allocations;
};

namespace {

template <typename T>
int dev_alloc(gsplat_ctx *c, T **out, size_t count, bool zero) {
    const size_t bytes = (count ? count : 1) * sizeof(T);
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    c->allocations.push_back(p);
    c->bytes_allocated += bytes;
    if (zero) HIP_TRY(hipMemsetAsync(p, 0, bytes, c->stream));
    *out = static_cast<T *>(p);
    return GSPLAT_OK;
}

int dev_free(gsplat_ctx *c, void *p, size_t bytes) {
    if (!p) return GSPLAT_OK;
    for (size_t i = 0; i < c->allocations.size(); ++i)
        if (c->allocations[i] == p) {
            c->allocations.erase(c->allocations.begin() + i);
            break;
        }
    c->bytes_allocated -= bytes;
    HIP_TRY(hipFree(p));
    return GSPLAT_OK;
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

int alloc_size_dependent(gsplat_ctx *c) {
    int rc;
    const size_t tpad = ((size_t)c->gx * c->gy + 1) & ~(size_t)1;  // tile_bounds, then the tile segments
    if ((rc = dev_alloc(c, &c->bounds, 2 * tpad, true))) return rc;
    c->segs = c->bounds + tpad;
    if ((rc = dev_alloc(c, &c->tile_staged, (size_t)c->gx * c->gy, true))) return rc;
    if ((rc = dev_alloc(c, &c->tile_big_list, (size_t)c->gx * c->gy, true))) return rc;
    if ((rc = dev_alloc(c, &c->image, (size_t)c->width * c->height, true))) return rc;
    return GSPLAT_OK;
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

}  // namespace

extern "C" {

int gsplat_create(const gsplat_config *config, gsplat_ctx **out_ctx) {
    if (!config || !out_ctx) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (config->struct_size != sizeof(gsplat_config)) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (config->width == 0 || config->height == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out_ctx = nullptr;
    const uint32_t gx = (config->width + TILE - 1) / TILE, gy = (config->height + TILE - 1) / TILE;
    // 16-bit tile ids (gsplat_projection.glsl:222) and 16-bit packed tile rectangles
    if ((uint64_t)gx * gy > 65536ull || gx > 65535u || gy > 65535u) return GSPLAT_ERR_OUT_OF_RANGE;
    const uint32_t factor = config->key_budget_factor ? config->key_budget_factor : 10u;
    const uint64_t capacity = (uint64_t)factor * config->max_splats;
    if (capacity >= 0xFFFFF000ull) return GSPLAT_ERR_OUT_OF_RANGE;  // pair indices are 32-bit
    if (config->sh_degree < -1 || config->sh_degree > 3) return GSPLAT_ERR_INVALID_ARGUMENT;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        return GSPLAT_ERR_NO_DEVICE;
    }
    int device = config->device_id;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return GSPLAT_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));

    gsplat_ctx *c = new (std::nothrow) gsplat_ctx();
    if (!c) return GSPLAT_ERR_OUT_OF_MEMORY;
    c->cfg = *config;
    c->cfg.key_budget_factor = factor;
    c->device = device;
    c->n = config->max_splats;
    c->capacity = capacity;
    c->width = config->width; c->height = config->height;
    c->gx = gx; c->gy = gy;

    int rc = GSPLAT_OK;
    do {
        if (config->stream) {
            c->stream = static_cast<hipStream_t>(config->stream);
        } else {
            hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
            if (e != hipSuccess) { rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); break; }
            c->own_stream = true;
        }
        hipError_t e = hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking);
        if (e != hipSuccess) { rc = hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); break; }
        if ((rc = apply_stripe(c, config->stripe_axis, config->stripe_begin, config->stripe_end))) break;

        const size_t n = c->n;
        // gaussian_splatting_rasterizer.gd:83-92, re-laid out as SoA
        if ((rc = dev_alloc(c, &c->scene.pos_time, n, true))) break;
        if ((rc = dev_alloc(c, &c->scene.cov_a, n, true))) break;
        if ((rc = dev_alloc(c, &c->scene.cov_b, n, true))) break;
        if ((rc = dev_alloc(c, &c->scene.sh_planes, n * SH_PLANES, true))) break;
        if ((rc = dev_alloc(c, &c->scene.sh, n * SH_BLOCK_F4, true))) break;
        if ((rc = dev_alloc(c, &c->culled, n * 3, true))) break;
        if ((rc = dev_alloc(c, &c->local_off, n, true))) break;
        if ((rc = dev_alloc(c, &c->counts, n, true))) break;
        if ((rc = dev_alloc(c, &c->depths, n, true))) break;
        if ((rc = dev_alloc(c, &c->rects, n, true))) break;
        c->num_proj_blocks = (uint32_t)((n + PROJ_BLOCK - 1) / PROJ_BLOCK);
        if ((rc = dev_alloc(c, &c->block_sums, (size_t)c->num_proj_blocks, true))) break;
        if ((rc = dev_alloc(c, &c->block_base, (size_t)c->num_proj_blocks, true))) break;
        if ((rc = dev_alloc(c, &c->big_list, (size_t)emit_big_list_entries(capacity), false))) break;
        {   // projection variant: project -> scan -> emit by default (measured faster at 6 M splats);
            // GSPLAT_PROJECT=fused selects the single-kernel variant with decoupled look-back for A/B runs
            const char *pv = getenv("GSPLAT_PROJECT");
            c->fused_projection = pv && strcmp(pv, "fused") == 0;
            if ((rc = dev_alloc(c, &c->chunk_status, (size_t)project_num_chunks(c->n), true))) break;
            if ((rc = dev_alloc(c, &c->chunk_info, (size_t)project_num_chunks(c->n), true))) break;
        }
        for (int h = 0; h < 2; ++h) {
            if ((rc = dev_alloc(c, &c->sort.keys[h], (size_t)capacity, false))) break;
            if ((rc = dev_alloc(c, &c->sort.values[h], (size_t)capacity, false))) break;
        }
        if (rc) break;
        if (config->flags & GSPLAT_FLAG_KEEP_EMITTED) {
            if ((rc = dev_alloc(c, &c->emit_keys, (size_t)capacity, false))) break;
            if ((rc = dev_alloc(c, &c->emit_values, (size_t)capacity, false))) break;
        }
        if ((rc = dev_alloc(c, &c->sort.part_hist, (size_t)sort_max_partitions(capacity) * 256, true))) break;
        if ((rc = dev_alloc(c, &c->sort.digit_base, 256, true))) break;
        {   // sort variant: reduce-then-scan by default (measured faster); GSPLAT_SORT=onesweep
            // selects the single-kernel-per-pass variant for A/B runs (needs capacity < 2^30 for its 30-bit counts)
            const char *cp = getenv("GSPLAT_COLOR");  // lazy | eager: pin where the SH colours are evaluated (A/B, tests)
            c->color_policy = cp && strcmp(cp, "lazy") == 0 ? 1 : (cp && strcmp(cp, "eager") == 0 ? 2 : 0);
            // three words the scan kernel posts to the host every frame (no copy, no synchronisation): the host reads
            // whatever is there when it sets up the next frame
            hipError_t he = hipHostMalloc(reinterpret_cast<void **>(&c->hint_host), 64, hipHostMallocMapped);
            if (he != hipSuccess) { rc = hip_fail(he, "hipHostMalloc", __FILE__, __LINE__); break; }
            memset(c->hint_host, 0, 64);
            he = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->hint_dev), c->hint_host, 0);
            if (he != hipSuccess) { rc = hip_fail(he, "hipHostGetDevicePointer", __FILE__, __LINE__); break; }
            const char *sp = getenv("GSPLAT_SORT_SMALL");  // A/B and tests: 0 = always 4096-key partitions
            c->sort.small_count = sp ? (uint32_t)strtoul(sp, nullptr, 10) : sort_small_count_default();
            if (c->sort.small_count > sort_small_count_default()) c->sort.small_count = sort_small_count_default();
            const char *sv = getenv("GSPLAT_SORT");
            c->sort.onesweep = (sv && strcmp(sv, "onesweep") == 0) && capacity < (1ull << 30);
            // GSPLAT_SORT=tile: tile-major variant (two global passes on the tile bits + per-tile depth sort,
            // tilesort.hip) — bit-identical, measured slower at every config of this round
            c->tile_major_sort = sv && strcmp(sv, "tile") == 0;
            if (c->sort.onesweep) {
                if ((rc = dev_alloc(c, &c->sort.global_hist, 4 * 256, true))) break;
                if ((rc = dev_alloc(c, &c->sort.status, (size_t)4 * sort_max_partitions(capacity) * 256, true))) break;
            }
        }
        if ((rc = dev_alloc(c, &c->pick, 1, true))) break;
        if ((rc = dev_alloc(c, &c->counters, 1, true))) break;
        c->sort.tickets = c->counters->tickets;
        c->sort.error_flag = &c->counters->sort_error;
        if ((rc = alloc_size_dependent(c))) break;
        for (int i = 0; i < 5; ++i) {
            e = hipEventCreate(&c->ev[i]);
            if (e != hipSuccess) { rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__); break; }
        }
        if (rc) break;
        for (int i = 0; i < 2; ++i) {
            e = hipEventCreate(&c->ev_tile[i]);
            if (e != hipSuccess) { rc = hip_fail(e, "hipEventCreate", __FILE__, __LINE__); break; }
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
    } while (0);
    if (rc != GSPLAT_OK) {
        gsplat_destroy(c);
        return rc;
    }
    *out_ctx = c;
    return GSPLAT_OK;
}

int gsplat_destroy(gsplat_ctx *c) {
    if (!c) return GSPLAT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->upload_stream) {
        (void)hipStreamSynchronize(c->upload_stream);
        (void)hipStreamDestroy(c->upload_stream);
    }
    for (void *p : c->allocations) (void)hipFree(p);
    if (c->hint_host) (void)hipHostFree(c->hint_host);
    for (int i = 0; i < 5; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 2; ++i)
        if (c->ev_tile[i]) (void)hipEventDestroy(c->ev_tile[i]);
    if (c->kt_events_created)
        for (int i = 0; i <= KernelTimer::MAX_MARKS; ++i) (void)hipEventDestroy(c->kt.ev[i]);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return GSPLAT_OK;
}

static int upload_common(gsplat_ctx *c, uint32_t first, uint32_t count, const float *src, int floats_per_item,
                         bool ply_rows, float load_time) {
    if (!c || (!src && count)) return GSPLAT_ERR_INVALID_ARGUMENT;
    if ((uint64_t)first + count > c->n) return GSPLAT_ERR_OUT_OF_RANGE;
    if (!count) return GSPLAT_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lock(c->upload_mutex);  // serialises the staging buffer; ranges may interleave
    const bool on_device = is_device_pointer(src);
    const uint32_t chunk_max = 1u << 20;  // 1 Mi items (~250 MB) per staging copy
    float *staging = nullptr;
    if (!on_device) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&staging),
                                      (size_t)(count < chunk_max ? count : chunk_max) * floats_per_item * 4));
    int rc = GSPLAT_OK;
    for (uint32_t done = 0; done < count && rc == GSPLAT_OK; done += chunk_max) {
        const uint32_t m = count - done < chunk_max ? count - done : chunk_max;
        const float *chunk_src = src + (size_t)done * floats_per_item;
        const float *d_src = chunk_src;
        if (!on_device) {
            hipError_t e = hipMemcpyAsync(staging, chunk_src, (size_t)m * floats_per_item * 4, hipMemcpyHostToDevice,
                                          c->upload_stream);
            if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__); break; }
            d_src = staging;
        }
        if (ply_rows)
            launch_upload_ply_rows(c->scene, c->n, first + done, m, d_src, load_time, &c->counters->sh_degree_max,
                                   c->finalized ? c->slot_of_id : nullptr, c->upload_stream);
        else
            launch_upload_records(c->scene, c->n, first + done, m, d_src, &c->counters->sh_degree_max,
                                  c->finalized ? c->slot_of_id : nullptr, c->upload_stream);
        hipError_t e = hipStreamSynchronize(c->upload_stream);
        if (e != hipSuccess) rc = hip_fail(e, "upload kernel", __FILE__, __LINE__);
    }
    if (c->finalized) c->bounds_dirty.store(true);  // the stored scene changed: block bounds are retaken by the next frame
    if (rc == GSPLAT_OK) {
        uint32_t deg = 0;
        hipError_t e = hipMemcpy(&deg, &c->counters->sh_degree_max, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
        else if ((int)deg > c->sh_degree_seen) c->sh_degree_seen = (int)deg;
    }
    if (staging) (void)hipFree(staging);
    return rc;
}

int gsplat_upload_splats(gsplat_ctx *c, uint32_t first, uint32_t count, const float *records60) {
    return upload_common(c, first, count, records60, GSPLAT_RECORD_FLOATS, false, 0.0f);
}

int gsplat_upload_ply_rows(gsplat_ctx *c, uint32_t first, uint32_t count, const float *rows62, float load_time) {
    return upload_common(c, first, count, rows62, GSPLAT_PLY_ROW_FLOATS, true, load_time);
}

int gsplat_finalize_scene(gsplat_ctx *c) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    if (c->finalized || c->n < 2) return GSPLAT_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::lock_guard<std::mutex> lock(c->upload_mutex);
    HIP_TRY(hipStreamSynchronize(c->upload_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t n = c->n;
    // 30-bit Morton code of the position inside the bounding box of the finite positions (host side: one-time,
    // load-time work like the reference's CPU swizzle, ply_file.gd:41-69)
    std::vector<float4> pos(n);
    HIP_TRY(hipMemcpy(pos.data(), c->scene.pos_time, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i) {
        const float p[3] = {pos[i].x, pos[i].y, pos[i].z};
        for (int a = 0; a < 3; ++a)
            if (std::isfinite(p[a])) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    auto spread = [](uint64_t v) {  // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FFull;
        v = (v | (v << 8)) & 0x0300F00Full;
        v = (v | (v << 4)) & 0x030C30C3ull;
        v = (v | (v << 2)) & 0x09249249ull;
        return v;
    };
    std::vector<uint64_t> order(n);  // (code << 32) | id: unique keys, plain sort is deterministic
    for (uint32_t i = 0; i < n; ++i) {
        const float p[3] = {pos[i].x, pos[i].y, pos[i].z};
        uint64_t code = 0;
        for (int a = 0; a < 3; ++a) {
            double t = 0.0;
            if (std::isfinite(p[a]) && hi[a] > lo[a]) t = ((double)p[a] - lo[a]) / ((double)hi[a] - lo[a]);
            uint64_t q = (uint64_t)(t * 1023.0);
            if (q > 1023) q = 1023;
            code |= spread(q) << a;
        }
        order[i] = (code << 32) | i;
    }
    std::sort(order.begin(), order.end());
    std::vector<uint32_t> id_of(n), slot_of(n);
    for (uint32_t slot = 0; slot < n; ++slot) {
        id_of[slot] = (uint32_t)(order[slot] & 0xFFFFFFFFull);
        slot_of[id_of[slot]] = slot;
    }
    int rc;
    if (!c->id_of_slot) {
        if ((rc = dev_alloc(c, &c->id_of_slot, n, false))) return rc;
        if ((rc = dev_alloc(c, &c->slot_of_id, n, false))) return rc;
    }
    HIP_TRY(hipMemcpy(c->id_of_slot, id_of.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->slot_of_id, slot_of.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    // permute the scene arrays through one temporary (the largest: 12 float4 of SH coefficients per splat)
    float4 *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (size_t)n * SH_BLOCK_F4 * sizeof(float4)));
    struct Arr { float4 *arr; uint32_t rec; };
    std::vector<Arr> arrays = {{c->scene.pos_time, 1u}, {c->scene.cov_a, 1u}, {c->scene.cov_b, 1u},
                               {c->scene.sh, (uint32_t)SH_BLOCK_F4}};
    for (int p = 0; p < SH_PLANES; ++p) arrays.push_back({c->scene.sh_planes + (size_t)p * n, 1u});
    for (const auto &a : arrays) {
        launch_permute_float4(a.arr, tmp, c->id_of_slot, n, a.rec, c->stream);
        hipError_t e = hipMemcpyAsync(a.arr, tmp, (size_t)n * a.rec * sizeof(float4), hipMemcpyDeviceToDevice,
                                      c->stream);
        if (e != hipSuccess) { (void)hipFree(tmp); return hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__); }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return hip_fail(e, "scene re-layout", __FILE__, __LINE__);
    if (!c->block_bounds) {
        if ((rc = dev_alloc(c, &c->block_bounds, (size_t)c->num_proj_blocks * 3, false))) return rc;
        if ((rc = dev_alloc(c, &c->block_skip, (size_t)c->num_proj_blocks, true))) return rc;
    }
    c->finalized = true;
    c->bounds_dirty.store(true);
    c->rendered = false;
    return GSPLAT_OK;
}

int gsplat_resize(gsplat_ctx *c, uint32_t width, uint32_t height) {
    if (!c || width == 0 || height == 0) return GSPLAT_ERR_INVALID_ARGUMENT;
    const uint32_t gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    if ((uint64_t)gx * gy > 65536ull || gx > 65535u || gy > 65535u) return GSPLAT_ERR_OUT_OF_RANGE;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc;
    if ((rc = dev_free(c, c->bounds, 2 * (((size_t)c->gx * c->gy + 1) & ~(size_t)1) * sizeof(uint2)))) return rc;
    c->bounds = nullptr;
    if ((rc = dev_free(c, c->tile_staged, (size_t)c->gx * c->gy * sizeof(uint32_t)))) return rc;
    c->tile_staged = nullptr;
    if ((rc = dev_free(c, c->tile_big_list, (size_t)c->gx * c->gy * sizeof(uint32_t)))) return rc;
    c->tile_big_list = nullptr;
    if ((rc = dev_free(c, c->image, (size_t)c->width * c->height * sizeof(float4)))) return rc;
    c->image = nullptr;
    c->width = width; c->height = height; c->gx = gx; c->gy = gy;
    c->cfg.width = width; c->cfg.height = height;
    if ((rc = alloc_size_dependent(c))) return rc;
    // a stripe is expressed in tiles of the old grid: fall back to the full frame
    if ((rc = apply_stripe(c, GSPLAT_STRIPE_NONE, 0, 0))) return rc;
    c->rendered = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return GSPLAT_OK;
}

int gsplat_set_stripe(gsplat_ctx *c, uint32_t axis, uint32_t b, uint32_t e) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    return apply_stripe(c, axis, b, e);
}

static bool is_sharded(const gsplat_ctx *c) {
    return c->sx0 > 0 || c->sy0 > 0 || c->sx1 < c->gx || c->sy1 < c->gy;
}

// First half of a frame: projection, key emission, sort.  stripe_cull: workgroups that cannot reach the context's
// stripe may be skipped too — then the "last tile" counter is stripe-local and the caller of render_back supplies the
// frame's (gsplat_render_end); without it only workgroups outside a frustum plane are skipped, which changes nothing.
static int render_front(gsplat_ctx *c, const gsplat_frame *frame, bool stripe_cull) {
    hipStream_t s = c->stream;
    FrameParams fp;
    fill_frame_params(c, frame, &fp);
    const bool timing = (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0;
    const int sh_degree = c->cfg.sh_degree >= 0 ? c->cfg.sh_degree : c->sh_degree_seen;
    const uint32_t tiles = c->gx * c->gy;
    const int sig_bits = sig_bits_for(tiles);
    KernelTimer *kt = c->kt.enabled ? &c->kt : nullptr;
    c->front_done = false;

    // Who evaluates the SH colours (gsplat_projection.glsl:198-201)?  Eager = the projection pass, for all V visible
    // splats, streaming 12 K bytes each; lazy = the compositor, for the D_c pairs it stages, gathering 12 K bytes each.
    // The two cost about the same per unit (0.21 ms / 5.9 M splats vs 0.11 ms / 3.0 M pairs at deg 3), so lazy pays
    // when D_c < V: heavy occlusion (6 M splats at 1080p: +9 % fps), not a 4K frame where every splat shows (-7 %).
    // V and D_c of the previous frames come from the words the scan kernel posts to host memory; 10 % hysteresis.
    bool lazy = c->last_lazy;
    if (sh_degree <= 0 || c->color_policy == 2) {
        lazy = false;  // band 0 only: 12 bytes per splat are cheaper to stream than to gather
    } else if (c->color_policy == 1) {
        lazy = true;
    } else {
        const volatile uint32_t *h = c->hint_host;
        const uint32_t v_prev = h[0], dc_prev = h[1], frames = h[2];
        if (frames < 2u) lazy = (uint64_t)c->n * 2u >= (uint64_t)c->width * c->height * 3u;  // no history: N >= 1.5 P
        else if ((uint64_t)dc_prev * 10u < (uint64_t)v_prev * 9u) lazy = true;
        else if ((uint64_t)dc_prev * 10u > (uint64_t)v_prev * 11u) lazy = false;
    }
    c->front_lazy = lazy;

    const float4 *block_bounds = nullptr;
    if ((c->cfg.flags & GSPLAT_FLAG_BLOCK_CULL) && c->finalized && c->block_bounds && !c->fused_projection) {
        if (c->bounds_dirty.exchange(false)) launch_block_bounds(c->scene, c->n, c->block_bounds, s);
        block_bounds = c->block_bounds;
        fp.cull_mode = stripe_cull ? 2u : 1u;
    }

    // gaussian_splatting_rasterizer.gd:127-128 clears the pair counter and tile_bounds with two buffer_clear calls;
    // here scan_blocks_kernel overwrites every per-frame counter and zeroes tile_bounds itself (no fill launches).
    // The fused-projection variant has no scan kernel and keeps the two clears.
    if (c->fused_projection) {
        HIP_TRY(hipMemsetAsync(c->counters, 0, offsetof(Counters, sort_error), s));
        HIP_TRY(hipMemsetAsync(&c->counters->big_count, 0, 2 * sizeof(uint32_t), s));
        HIP_TRY(hipMemsetAsync(c->bounds, 0, 2 * (((size_t)tiles + 1) & ~(size_t)1) * sizeof(uint2), s));
    }

    if (timing) HIP_TRY(hipEventRecord(c->ev[0], s));  // 'Start'
    c->kt.begin(s);
    if (c->fused_projection) {
        launch_project_emit(c->scene, c->n, fp, lazy ? -1 : sh_degree, c->culled, c->counts, c->chunk_status,
                            &c->counters->proj_ticket, c->chunk_info, c->capacity, c->sort.keys[0], c->sort.values[0],
                            &c->counters->total_emitted, &c->counters->d_sorted, &c->counters->overflow,
                            &c->counters->visible, &c->counters->frame_last_tile_plus1, &c->counters->sort_error, s);
        if (kt) kt->mark(GSPLAT_KERNEL_PROJECT);
    } else {
        launch_project(c->scene, c->n, fp, lazy ? -1 : sh_degree, c->culled, c->local_off, c->counts, c->rects, c->depths,
                       c->block_sums, block_bounds, c->block_skip, s);
        if (kt) kt->mark(GSPLAT_KERNEL_PROJECT);
        launch_scan_blocks(c->block_sums, c->num_proj_blocks, c->block_base, c->capacity,
                           &c->counters->total_emitted, &c->counters->d_sorted, &c->counters->overflow,
                           &c->counters->visible, &c->counters->frame_last_tile_plus1, c->bounds,
                           2u * ((tiles + 1u) & ~1u), &c->counters->big_count, c->tile_staged, tiles, c->hint_dev, s);
        if (kt) kt->mark(GSPLAT_KERNEL_SCAN);
        launch_emit(c->n, fp, c->local_off, c->counts, c->rects, c->depths, c->block_sums, c->block_base,
                    c->capacity, c->sort.keys[0], c->sort.values[0], &c->counters->big_count, c->big_list, s);
        if (kt) kt->mark(GSPLAT_KERNEL_EMIT);
    }
    if (c->emit_keys) {
        HIP_TRY(hipMemcpyAsync(c->emit_keys, c->sort.keys[0], (size_t)c->capacity * 4, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->emit_values, c->sort.values[0], (size_t)c->capacity * 4, hipMemcpyDeviceToDevice, s));
    }
    if (timing) HIP_TRY(hipEventRecord(c->ev[1], s));  // 'Projection'
    // tile-major: only the tile bits are sorted globally here; render_back sorts every tile's segment by depth
    const bool tile_major = c->tile_major_sort && !c->sort.onesweep;
    c->sorted_index = launch_sort_pairs(c->sort, &c->counters->d_sorted, c->capacity, sig_bits, s, kt,
                                        tile_major ? 16 : 0);
    if (timing) HIP_TRY(hipEventRecord(c->ev[2], s));  // 'Sort'
    HIP_TRY(hipGetLastError());
    c->front_fp = fp;
    c->front_sig_bits = sig_bits;
    c->front_sh_degree = sh_degree;
    c->front_done = true;
    c->rendered = false;
    return GSPLAT_OK;
}

// Second half: tile ranges + compositor.  last_tile_dev: device word holding the frame's highest populated tile + 1
// (nullptr = this context's own counter).
static int render_back(gsplat_ctx *c, float4 *target, uint32_t pitch, uint32_t ox, uint32_t oy,
                       const uint32_t *last_tile_dev) {
    if (!c->front_done) return GSPLAT_ERR_INVALID_ARGUMENT;
    hipStream_t s = c->stream;
    const FrameParams &fp = c->front_fp;
    const bool timing = (c->cfg.flags & GSPLAT_FLAG_TIMING) != 0;
    const uint32_t tiles = c->gx * c->gy;
    KernelTimer *kt = c->kt.enabled ? &c->kt : nullptr;
    const bool tile_major = c->tile_major_sort && !c->sort.onesweep;
    const bool fix_last = (c->cfg.flags & GSPLAT_FLAG_FIX_LAST_TILE) != 0;
    const uint32_t *last_tile = last_tile_dev ? last_tile_dev : &c->counters->frame_last_tile_plus1;
    const int si = c->sorted_index;
    c->values_index = c->finalized ? (si ^ 1) : si;
    if (!tile_major) {
        launch_boundaries(c->sort.keys[si], &c->counters->d_sorted, tiles, c->bounds, nullptr, fix_last, is_sharded(c),
                          last_tile, c->finalized ? c->sort.values[si] : nullptr,
                          c->finalized ? c->sort.values[c->values_index] : nullptr, c->id_of_slot, s);
        if (kt) kt->mark(GSPLAT_KERNEL_BOUNDARIES);
        if (timing) HIP_TRY(hipEventRecord(c->ev[3], s));  // 'Boundaries'
        c->tile_timing_valid = false;
    } else {
        c->tile_timing_valid = timing;
        // the pairs are grouped by tile (emission order inside a tile): tile ranges first — they only look at the tile
        // bits —, then every tile's segment is sorted by depth in place
        launch_boundaries(c->sort.keys[si], &c->counters->d_sorted, tiles, c->bounds, c->segs, fix_last, is_sharded(c),
                          last_tile, nullptr, nullptr, nullptr, s);
        if (kt) kt->mark(GSPLAT_KERNEL_BOUNDARIES);
        if (timing) HIP_TRY(hipEventRecord(c->ev_tile[0], s));
        if (launch_tile_depth_sort(c->sort.keys[si], c->sort.values[si], c->sort.keys[si ^ 1], c->sort.values[si ^ 1],
                                   c->segs, tiles, &c->counters->d_sorted, &c->counters->tile_big_count,
                                   c->tile_big_list, s) != 0)
            return hip_fail(hipGetLastError(), "tile sort LDS attribute", __FILE__, __LINE__);
        if (kt) kt->mark(GSPLAT_KERNEL_TILE_SORT);
        if (timing) HIP_TRY(hipEventRecord(c->ev_tile[1], s));
        if (c->finalized) {  // equal keys back to ascending splat id (the pass re-derives the same tile ranges)
            launch_boundaries(c->sort.keys[si], &c->counters->d_sorted, tiles, c->bounds, nullptr, fix_last,
                              is_sharded(c), last_tile, c->sort.values[si], c->sort.values[c->values_index],
                              c->id_of_slot, s);
            if (kt) kt->mark(GSPLAT_KERNEL_BOUNDARIES);
        }
        if (timing) HIP_TRY(hipEventRecord(c->ev[3], s));  // 'Boundaries' (minus the depth sort, see gsplat_get_stats)
    }
    launch_render(c->culled, c->scene.sh, c->front_lazy ? c->front_sh_degree : -1, c->sort.values[c->values_index],
                  c->bounds, fp, target,
                  pitch, ox, oy, c->pick,
                  c->tile_staged, (c->cfg.flags & GSPLAT_FLAG_FAST_EXP) != 0, s);
    if (kt) kt->mark(GSPLAT_KERNEL_RENDER);
    if (timing) HIP_TRY(hipEventRecord(c->ev[4], s));  // 'Render'
    HIP_TRY(hipGetLastError());
    c->timing_valid = timing;
    c->last_sig_bits = c->front_sig_bits;
    c->last_sh_degree = c->front_sh_degree;
    c->last_fp = c->front_fp;
    c->last_lazy = c->front_lazy;
    c->front_done = false;
    c->rendered = true;
    return GSPLAT_OK;
}

static int render_impl(gsplat_ctx *c, const gsplat_frame *frame, float4 *target, uint32_t pitch, uint32_t ox,
                       uint32_t oy) {
    // one call, no exchange: a stripe context may only skip what cannot change its "last tile" counter
    const int rc = render_front(c, frame, /*stripe_cull=*/!is_sharded(c));
    if (rc != GSPLAT_OK) return rc;
    return render_back(c, target, pitch, ox, oy, nullptr);
}

int gsplat_render(gsplat_ctx *c, const gsplat_frame *frame, float *rgba_out) {
    if (!c || !frame) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    float4 *target = c->image;
    bool copy_to_host = false;
    if (rgba_out) {
        if (is_device_pointer(rgba_out)) target = reinterpret_cast<float4 *>(rgba_out);
        else copy_to_host = true;
    }
    const int rc = render_impl(c, frame, target, c->width, 0, 0);
    if (rc != GSPLAT_OK) return rc;
    if (copy_to_host) {
        HIP_TRY(hipMemcpyAsync(rgba_out, c->image, (size_t)c->width * c->height * sizeof(float4),
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
    return render_impl(c, frame, reinterpret_cast<float4 *>(device_out), pitch_px, origin_x, origin_y);
}

int gsplat_render_begin(gsplat_ctx *c, const gsplat_frame *frame, uint32_t *last_tile_out_device) {
    if (!c || !frame) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    const int rc = render_front(c, frame, /*stripe_cull=*/true);
    if (rc != GSPLAT_OK) return rc;
    if (last_tile_out_device)
        HIP_TRY(hipMemcpyAsync(last_tile_out_device, &c->counters->frame_last_tile_plus1, sizeof(uint32_t),
                               hipMemcpyDeviceToDevice, c->stream));
    return GSPLAT_OK;
}

int gsplat_render_end(gsplat_ctx *c, float *device_out, uint32_t pitch_px, uint32_t origin_x, uint32_t origin_y,
                      const uint32_t *frame_last_tile_device) {
    if (!c) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    if (!device_out) return render_back(c, c->image, c->width, 0, 0, frame_last_tile_device);
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
    if (tile_id >= c->gx * c->gy) return GSPLAT_ERR_OUT_OF_RANGE;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    FrameParams fp;
    fill_frame_params(c, frame, &fp);
    fp.target_tile = tile_id;
    // gaussian_splatting_rasterizer.gd:166-168 re-runs the whole compositor; only the target tile can write
    // the pick record, so a 1x1 grid on that tile gives the same 16 bytes.
    const uint32_t tx = tile_id % c->gx, ty = tile_id / c->gx;
    if (tx < c->sx0 || tx >= c->sx1 || ty < c->sy0 || ty >= c->sy1) return GSPLAT_ERR_OUT_OF_RANGE;
    fp.sx0 = tx; fp.sx1 = tx + 1; fp.sy0 = ty; fp.sy1 = ty + 1;
    HIP_TRY(hipMemsetAsync(c->pick, 0, sizeof(float4), s));  // no stale hits
    launch_render(c->culled, c->scene.sh, c->last_lazy ? c->last_sh_degree : -1, c->sort.values[c->values_index],
                  c->bounds, fp, c->image,
                  c->width, 0, 0, c->pick,
                  nullptr, (c->cfg.flags & GSPLAT_FLAG_FAST_EXP) != 0, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_xyzn, c->pick, sizeof(float4), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GSPLAT_OK;
}

int gsplat_get_stats(gsplat_ctx *c, gsplat_stats *out) {
    if (!c || !out) return GSPLAT_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof h, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->num_splats = c->n;
    out->num_visible = h.visible;
    out->num_emitted = h.total_emitted;
    out->num_sorted = h.d_sorted;
    {   // D_c = sum over this context's tiles of the pairs the compositor staged
        std::vector<uint32_t> staged((size_t)c->gx * c->gy);
        HIP_TRY(hipMemcpy(staged.data(), c->tile_staged, staged.size() * 4, hipMemcpyDeviceToHost));
        uint64_t dc = 0;
        for (uint32_t ty = c->sy0; ty < c->sy1; ++ty)
            for (uint32_t tx = c->sx0; tx < c->sx1; ++tx) dc += staged[(size_t)ty * c->gx + tx];
        out->num_composited = c->rendered ? dc : 0;
    }
    out->capacity = c->capacity;
    out->overflow = (int32_t)h.overflow;
    if (h.sort_error) {
        snprintf(g_last_error, sizeof g_last_error, "radix sort look-back timed out");
        return GSPLAT_ERR_HIP;
    }
    out->sort_passes = sort_num_passes(c->last_sig_bits);
    out->sh_degree = c->last_sh_degree;
    out->lazy_colors = c->last_lazy ? 1 : 0;
    out->bytes_allocated = c->bytes_allocated;
    if (c->timing_valid) {
        HIP_TRY(hipEventElapsedTime(&out->ms_projection, c->ev[0], c->ev[1]));
        HIP_TRY(hipEventElapsedTime(&out->ms_sort, c->ev[1], c->ev[2]));
        HIP_TRY(hipEventElapsedTime(&out->ms_boundaries, c->ev[2], c->ev[3]));
        if (c->tile_timing_valid) {  // the per-tile depth sort runs between the two boundary marks: it is sort time
            float ms_tile = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms_tile, c->ev_tile[0], c->ev_tile[1]));
            out->ms_sort += ms_tile;
            out->ms_boundaries -= ms_tile;
        }
        HIP_TRY(hipEventElapsedTime(&out->ms_render, c->ev[3], c->ev[4]));
        HIP_TRY(hipEventElapsedTime(&out->ms_total, c->ev[0], c->ev[4]));
    }
    if (c->kt.enabled && c->rendered) {
        for (int i = 0; i < c->kt.count; ++i) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, c->kt.ev[i], c->kt.ev[i + 1]));
            out->ms_kernel[c->kt.cls[i]] += ms;
            out->launches_kernel[c->kt.cls[i]] += 1;
        }
    }
    // SURVEY.md §8(d) algorithmic bytes; K = coefficients per channel actually evaluated
    const uint64_t N = c->n, V = h.visible, D = h.d_sorted;
    const uint64_t K = (uint64_t)(c->last_sh_degree + 1) * (c->last_sh_degree + 1);
    const uint64_t T = (uint64_t)c->gx * c->gy, P = (uint64_t)c->width * c->height;
    out->algorithmic_bytes[0] = 16 * N + (28 + 12 * K) * V + 48 * V + 8 * D;
    out->algorithmic_bytes[1] = 4 * D + (uint64_t)sort_num_passes(c->last_sig_bits) * 16 * D;
    out->algorithmic_bytes[2] = 4 * D + 8 * T;
    out->algorithmic_bytes[3] = 40 * D + 16 * P;
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
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->counters, sizeof h, hipMemcpyDeviceToHost));
    const void *src = nullptr;
    size_t avail = 0;
    float *tmp = nullptr;
    // a re-laid-out scene keeps per-splat arrays in storage order and slot numbers in the value arrays: the taps
    // present everything in splat-id terms, like a context that was never finalized
    auto mapped_u32 = [&](const uint32_t *srcp, const uint32_t *index, size_t count) -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), (count ? count : 4) * 4));
        launch_gather_u32(srcp, reinterpret_cast<uint32_t *>(tmp), index, (uint32_t)count, c->stream);
        HIP_TRY(hipStreamSynchronize(c->stream));
        src = tmp;
        avail = count * 4;
        return GSPLAT_OK;
    };
    switch (which) {
        case GSPLAT_DEBUG_CULLED:
            avail = (size_t)c->n * 48;
            // the frame evaluates colours only for the splats it stages; the tap shows the reference's full record
            if (c->rendered && c->last_lazy) {
                launch_fill_colors(c->culled, c->scene.sh, c->last_sh_degree, c->counts, c->n, c->last_fp, c->stream);
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            if (c->finalized) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), avail ? avail : 16));
                launch_gather_raster(c->culled, reinterpret_cast<float4 *>(tmp), c->slot_of_id, c->n, c->stream);
                HIP_TRY(hipStreamSynchronize(c->stream));
                src = tmp;
            } else {
                src = c->culled;
            }
            break;
        case GSPLAT_DEBUG_KEYS_SORTED: src = c->sort.keys[c->sorted_index]; avail = (size_t)h.d_sorted * 4; break;
        case GSPLAT_DEBUG_VALUES_SORTED:
            if (c->finalized) {  // value v is a slot: present id_of_slot[v]
                const int rc = mapped_u32(c->id_of_slot, c->sort.values[c->values_index], h.d_sorted);
                if (rc) return rc;
            } else {
                src = c->sort.values[c->values_index];
                avail = (size_t)h.d_sorted * 4;
            }
            break;
        case GSPLAT_DEBUG_TILE_BOUNDS: src = c->bounds; avail = (size_t)c->gx * c->gy * 8; break;
        case GSPLAT_DEBUG_KEYS_EMITTED:
        case GSPLAT_DEBUG_VALUES_EMITTED:
            // ping-pong half 0 is overwritten by the second sort pass: needs GSPLAT_FLAG_KEEP_EMITTED.  After
            // gsplat_finalize_scene the emission order is the storage order, not ascending splat id.
            if (!c->emit_keys) return GSPLAT_ERR_UNSUPPORTED;
            if (which == GSPLAT_DEBUG_VALUES_EMITTED && c->finalized) {
                const int rc = mapped_u32(c->id_of_slot, c->emit_values, h.d_sorted);
                if (rc) return rc;
            } else {
                src = which == GSPLAT_DEBUG_KEYS_EMITTED ? c->emit_keys : c->emit_values;
                avail = (size_t)h.d_sorted * 4;
            }
            break;
        case GSPLAT_DEBUG_TILE_COUNTS:
            if (c->finalized) {
                const int rc = mapped_u32(c->counts, c->slot_of_id, c->n);
                if (rc) return rc;
            } else {
                src = c->counts;
                avail = (size_t)c->n * 4;
            }
            break;
        case GSPLAT_DEBUG_TILE_STAGED: src = c->tile_staged; avail = (size_t)c->gx * c->gy * 4; break;
        case GSPLAT_DEBUG_BLOCK_SUMS: src = c->block_sums; avail = (size_t)c->num_proj_blocks * 16; break;
        case GSPLAT_DEBUG_IMAGE: src = c->image; avail = (size_t)c->width * c->height * 16; break;
        case GSPLAT_DEBUG_RECORDS: {
            avail = (size_t)c->n * 240;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), avail ? avail : 16));
            launch_gather_records(c->scene, c->n, tmp, c->finalized ? c->slot_of_id : nullptr, c->stream);
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { (void)hipFree(tmp); return hip_fail(e, "gather", __FILE__, __LINE__); }
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
    if (tmp) (void)hipFree(tmp);
    if (bytes_written) *bytes_written = nbytes;
    return rc;
}

int gsplat_image_device_ptr(gsplat_ctx *c, float **out_ptr) {
    if (!c || !out_ptr) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out_ptr = reinterpret_cast<float *>(c->image);
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
