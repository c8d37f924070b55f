// Multi-GPU frames behind the C ABI (include/gsplat.h: gsplat_group_*): tile-stripe sharding of one frame over the GPUs
// of a node, the 4-byte all-reduce of the frame's last tile and the all-gather-v of the finished stripes issued by the
// library itself on the members' streams.  RCCL (xGMI) is loaded on first use with dlopen — libgsplat_hip.so has no
// link-time dependency on it and a single-GPU host never loads it.  No reference counterpart (the reference is
// single-GPU); protocol and stripe geometry are those of godotgaussiansplatting_amd/distributed.py.
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/gsplat.h"
#include "gsplat_internal.h"

using namespace gsplat;

static_assert(sizeof(ncclUniqueId) == GSPLAT_GROUP_ID_BYTES, "gsplat.h: GSPLAT_GROUP_ID_BYTES");

namespace {

// the handful of RCCL entry points the exchange uses
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return GSPLAT_OK;
    // a copy that is already in the process wins (a host that also uses torch.distributed has torch's own librccl, built
    // against the HIP runtime the process runs on); then GSPLAT_RCCL_LIB, then the system's
    const char *env = getenv("GSPLAT_RCCL_LIB");
    const char *names[] = {env, "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *n : {"librccl.so", "librccl.so.1"})
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
        if (!h && n && *n) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) return set_last_error("librccl not found (set GSPLAT_RCCL_LIB)", GSPLAT_ERR_UNSUPPORTED);
    Rccl r;
    r.handle = h;
#define GSPLAT_SYM(field, name)                                                                  \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                               \
    if (!r.field) return set_last_error("librccl lacks " name, GSPLAT_ERR_UNSUPPORTED)
    GSPLAT_SYM(GetUniqueId, "ncclGetUniqueId");
    GSPLAT_SYM(CommInitRank, "ncclCommInitRank");
    GSPLAT_SYM(CommInitAll, "ncclCommInitAll");
    GSPLAT_SYM(CommDestroy, "ncclCommDestroy");
    GSPLAT_SYM(GroupStart, "ncclGroupStart");
    GSPLAT_SYM(GroupEnd, "ncclGroupEnd");
    GSPLAT_SYM(AllReduce, "ncclAllReduce");
    GSPLAT_SYM(Broadcast, "ncclBroadcast");
    GSPLAT_SYM(Send, "ncclSend");
    GSPLAT_SYM(Recv, "ncclRecv");
    GSPLAT_SYM(GetErrorString, "ncclGetErrorString");
#undef GSPLAT_SYM
    g_rccl = r;
    return GSPLAT_OK;
}

int nccl_fail(ncclResult_t r, const char *what) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return set_last_error(buf, GSPLAT_ERR_HIP);
}
#define NCCL_TRY(expr)                                   \
    do {                                                 \
        ncclResult_t _r = (expr);                        \
        if (_r != ncclSuccess) return nccl_fail(_r, #expr); \
    } while (0)
#define HIP_TRY_G(expr)                                                          \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return set_last_error(hipGetErrorString(_e), GSPLAT_ERR_HIP); \
    } while (0)

// What travels: by default the three colour channels of a pixel, 12 bytes — alpha is the constant 1.0 of
// gsplat_render.glsl:101 and is rebuilt on arrival — through a contiguous, stripe-major staging buffer
// (member r's stripe at stage_off[r]; inside a stripe row-major over the stripe's own rectangle).  At 4K an 8-GPU
// all-gather-v moves 7/8 of 133 MB INTO every GPU per frame with RGBA32F: xGMI, not the kernels, then bounds the frame
// rate; 12 bytes per pixel is 25 % less.  GSPLAT_GROUP_PIXELS=rgba keeps the 16-byte form (row stripes: sent and received
// in place in the row-major image, no staging; column stripes: packed as float4).
// One pixel of the stripe rectangle [x0, x0 + w) x [y0, y0 + h) per lane.
// (blockIdx.y = frame of a batch: image k at image + k * image_stride_px, its packed stripe right behind frame k - 1's)
template <bool RGB>
__global__ __launch_bounds__(256) void pack_stripe_kernel(const float4 *__restrict__ image, uint32_t pitch, uint32_t x0,
                                                          uint32_t y0, uint32_t w, uint32_t h, float *__restrict__ packed,
                                                          uint32_t image_stride_px = 0) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= w * h) return;
    image += (size_t)blockIdx.y * image_stride_px;
    packed += (size_t)blockIdx.y * w * h * (RGB ? 3u : 4u);
    const float4 px = image[(size_t)(y0 + i / w) * pitch + x0 + i % w];
    if (RGB) {
        packed[3 * (size_t)i + 0] = px.x; packed[3 * (size_t)i + 1] = px.y; packed[3 * (size_t)i + 2] = px.z;
    } else {
        reinterpret_cast<float4 *>(packed)[i] = px;
    }
}
// Every stripe but `skip` (the member's own, already in place) from the staging buffer into the row-major image: ONE launch
// for all peers.  rects: per member {x0, y0, w, h}; first[r]: pixels of the stripes before r (stage_off in pixels).
struct UnpackArgs {
    uint32_t x0[8], y0[8], w[8], h[8], first[9];  // (groups of up to 8 members take this kernel; larger: one launch per peer)
    int n, skip;
};
// (frames > 1, a batch: blockIdx.y = frame; member r's `frames` stripes sit one after the other at frames * first[r])
template <bool RGB>
__global__ __launch_bounds__(256) void unpack_stripes_kernel(const float *__restrict__ packed, uint32_t pitch, UnpackArgs a,
                                                             float4 *__restrict__ image, uint32_t frames = 1,
                                                             uint32_t image_stride_px = 0) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.first[a.n]) return;
    int r = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) r += (k < a.n && i >= a.first[k]) ? 1 : 0;
    if (r == a.skip) return;
    const uint32_t j = i - a.first[r], w = a.w[r];
    // pixel j of member r's stripe of frame blockIdx.y
    const size_t at = (size_t)frames * a.first[r] + (size_t)blockIdx.y * (a.first[r + 1] - a.first[r]) + j;
    float4 px;
    if (RGB) px = make_float4(packed[3 * at + 0], packed[3 * at + 1], packed[3 * at + 2], 1.0f);
    else px = reinterpret_cast<const float4 *>(packed)[at];
    image[(size_t)blockIdx.y * image_stride_px + (size_t)(a.y0[r] + j / w) * pitch + a.x0[r] + j % w] = px;
}

}  // namespace

namespace gsplat {
// (declared where it is used: api.hip's read-back ring packs whole frames with the stripes' kernel)
void launch_pack_rgb(const float4 *image, uint32_t pitch, uint32_t w, uint32_t h, float *packed_rgb, hipStream_t s) {
    const uint32_t px = w * h;
    if (px == 0u) return;
    hipLaunchKernelGGL(pack_stripe_kernel<true>, dim3((px + 255u) / 256u), dim3(256), 0, s, image, pitch, 0u, 0u, w, h, packed_rgb);
}
}  // namespace gsplat

namespace {

struct Member {
    gsplat_ctx *ctx = nullptr;
    int rank = 0;
    ncclComm_t comm = nullptr;
    uint32_t *last_tile = nullptr;   // device words: [0] this member's / the frame's highest populated tile + 1;
                                     // [4..7] scratch of the agreement at creation
    float *staging = nullptr;        // every member's stripe, packed (12 or 16 bytes per pixel), one after the other
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool joined = false;             // ctx_join_group succeeded: gsplat_group_destroy hands the context back
};

}  // namespace

struct gsplat_group {
    int world = 0;
    uint32_t axis = GSPLAT_STRIPE_ROWS;
    uint32_t width = 0, height = 0, gx = 0, gy = 0;
    std::vector<Member> members;      // the LOCAL members (one per process in the rank form, all of them in the local form)
    std::vector<uint32_t> cuts;       // world + 1 stripe boundaries in tiles
    std::vector<size_t> stage_off;    // PIXEL offset of member r's stripe in the staging buffer
    bool p2p = true;                  // the all-gather-v as direct sends / receives (GSPLAT_GROUP_GATHER=broadcast: grouped broadcasts)
    bool rgb = true;                  // 12 bytes per pixel travel (GSPLAT_GROUP_PIXELS=rgba: 16)
    bool staged = true;               // the stripes travel through the staging buffer (false: rgba row stripes, in place)
    bool exchange = false;            // every frame carries the 4-byte all-reduce of the frame's last tile: decided ONCE, at
                                      // creation, from all ranks' states (MAX) — never again per frame from local state
};

namespace {

uint32_t px_lo(const gsplat_group *g, int r) {
    const uint32_t lim = g->axis == GSPLAT_STRIPE_COLUMNS ? g->width : g->height;
    const uint32_t v = g->cuts[r] * GSPLAT_TILE_SIZE;
    return v < lim ? v : lim;
}
uint32_t px_hi(const gsplat_group *g, int r) {
    const uint32_t lim = g->axis == GSPLAT_STRIPE_COLUMNS ? g->width : g->height;
    const uint32_t v = g->cuts[r + 1] * GSPLAT_TILE_SIZE;
    return v < lim ? v : lim;
}
// member r's stripe as a pixel rectangle
struct Rect { uint32_t x0, y0, w, h; };
Rect stripe_rect(const gsplat_group *g, int r) {
    const uint32_t lo = px_lo(g, r), hi = px_hi(g, r);
    if (g->axis == GSPLAT_STRIPE_COLUMNS) return Rect{lo, 0u, hi - lo, g->height};
    return Rect{0u, lo, g->width, hi - lo};
}

int apply_cuts(gsplat_group *g) {
    g->stage_off.assign(g->world + 1, 0);
    for (int r = 0; r < g->world; ++r) {
        const Rect rc = stripe_rect(g, r);
        g->stage_off[r + 1] = g->stage_off[r] + (size_t)rc.w * rc.h;
    }
    for (Member &m : g->members) {
        const int rc = gsplat_set_stripe(m.ctx, g->axis, g->cuts[m.rank], g->cuts[m.rank + 1]);
        if (rc != GSPLAT_OK) return rc;
    }
    return GSPLAT_OK;
}

// Does every frame of this group carry the 4-byte all-reduce?  Only a member that may skip whole blocks of the scene
// against its stripe (GSPLAT_FLAG_BLOCK_CULL on a finalized scene) cannot know the frame's highest populated tile by
// itself.  Round 4 decided this per frame and per rank from the rank's own state: a rank that finalized its scene a
// frame earlier than its peers, or was created with other flags, entered ncclAllReduce while the peers went on to their
// sends and receives — a hang for good.  Now the ranks agree ONCE, here (collective: every rank is inside
// gsplat_group_create anyway): MAX over the ranks of {wants, does not want}; the group exchanges if ANY rank wants to,
// and a member whose own state differs later (a scene finalized after the group was made) simply renders without the
// stripe part of the culling (ctx_render_begin's stripe_cull = the group's word, never the context's).
int agree_on_exchange(gsplat_group *g) {
    bool wants = false;
    for (Member &m : g->members) wants = wants || ctx_view(m.ctx).stripe_cull;
    g->exchange = wants;
    // ... and the members of one frame must resolve equal keys the same way (GSPLAT_FLAG_TIES_STORAGE_ORDER, gsplat.h): with
    // mixed flags the two sides of a stripe seam would composite a run of equal keys in different orders — a frame no
    // single context renders.  Checked here, where every rank is present: local members directly, ranks through the same
    // creation-time all-reduces (MAX of the bit and of its complement: both set = the ranks differ).
    const bool ties = ctx_view(g->members[0].ctx).ties_storage;
    for (Member &m : g->members)
        if (ctx_view(m.ctx).ties_storage != ties)
            return set_last_error("gsplat_group_create: the members differ in GSPLAT_FLAG_TIES_STORAGE_ORDER",
                                  GSPLAT_ERR_INVALID_ARGUMENT);
    if (g->world <= 1 || g->members.size() != 1) return GSPLAT_OK;  // (local form: every member is here, `wants` is the OR)
    Member &m = g->members[0];
    const CtxView v = ctx_view(m.ctx);
    const uint32_t mine[4] = {wants ? 1u : 0u, wants ? 0u : 1u, ties ? 1u : 0u, ties ? 0u : 1u};
    uint32_t all[4] = {0u, 0u, 0u, 0u};
    HIP_TRY_G(hipSetDevice(v.device));
    HIP_TRY_G(hipMemcpyAsync(m.last_tile + 4, mine, sizeof mine, hipMemcpyHostToDevice, v.stream));
    // (single-word all-reduces: the library — and the test suite's stand-in for RCCL — issue no other shape)
    for (int k = 0; k < 4; ++k)
        NCCL_TRY(g_rccl.AllReduce(m.last_tile + 4 + k, m.last_tile + 4 + k, 1, ncclUint32, ncclMax, m.comm, v.stream));
    HIP_TRY_G(hipMemcpyAsync(all, m.last_tile + 4, sizeof all, hipMemcpyDeviceToHost, v.stream));
    HIP_TRY_G(hipStreamSynchronize(v.stream));
    g->exchange = all[0] != 0u;
    if (all[2] != 0u && all[3] != 0u)  // (every rank sees the same words: all of them fail, none is left in a collective)
        return set_last_error("gsplat_group_create: the ranks' members differ in GSPLAT_FLAG_TIES_STORAGE_ORDER",
                              GSPLAT_ERR_INVALID_ARGUMENT);
    if (all[0] != 0u && all[1] != 0u && getenv("GSPLAT_GROUP_QUIET") == nullptr)
        fprintf(stderr, "gsplat_group_create: the ranks' members differ in GSPLAT_FLAG_BLOCK_CULL / gsplat_finalize_scene; "
                        "every frame will carry the last-tile exchange and rank %d %s\n", m.rank,
                wants ? "culls by stripe" : "does not cull");
    return GSPLAT_OK;
}

int finish_create(gsplat_group *g, uint32_t axis) {
    if (axis != GSPLAT_STRIPE_ROWS && axis != GSPLAT_STRIPE_COLUMNS) return GSPLAT_ERR_INVALID_ARGUMENT;
    const CtxView v0 = ctx_view(g->members[0].ctx);
    const char *gm = getenv("GSPLAT_GROUP_GATHER");
    g->p2p = !(gm && !strcmp(gm, "broadcast"));
    const char *pm = getenv("GSPLAT_GROUP_PIXELS");
    g->rgb = !(pm && !strcmp(pm, "rgba"));
    g->axis = axis;
    g->staged = g->rgb || axis == GSPLAT_STRIPE_COLUMNS;
    g->width = v0.width; g->height = v0.height; g->gx = v0.gx; g->gy = v0.gy;
    const uint32_t extent = axis == GSPLAT_STRIPE_COLUMNS ? g->gx : g->gy;
    g->cuts.resize(g->world + 1);
    for (int r = 0; r <= g->world; ++r) g->cuts[r] = (uint32_t)(((uint64_t)r * extent) / (uint32_t)g->world);
    for (Member &m : g->members) {
        const CtxView v = ctx_view(m.ctx);
        if (v.width != g->width || v.height != g->height) return GSPLAT_ERR_INVALID_ARGUMENT;
        // the group caches the member's size and pointer: from here to gsplat_group_destroy the context refuses
        // gsplat_resize and gsplat_destroy (api.hip), and it can be in one group only
        if (!ctx_join_group(m.ctx, g)) return set_last_error("the context is already in a group", GSPLAT_ERR_INVALID_ARGUMENT);
        m.joined = true;
        HIP_TRY_G(hipSetDevice(v.device));
        HIP_TRY_G(hipMalloc(reinterpret_cast<void **>(&m.last_tile), 64));
        HIP_TRY_G(hipMemset(m.last_tile, 0, 64));
        // (a batch context's member exchanges up to its batch of frames at once: gsplat_group_render_batch)
        const uint32_t cap = ctx_batch_capacity(m.ctx);
        if ((g->staged || cap > 1u) && g->world > 1)
            HIP_TRY_G(hipMalloc(reinterpret_cast<void **>(&m.staging),
                                (size_t)g->width * g->height * (g->rgb ? 3 : 4) * sizeof(float) * cap));
        HIP_TRY_G(hipEventCreate(&m.t0));
        HIP_TRY_G(hipEventCreate(&m.t1));
    }
    int rc = apply_cuts(g);
    if (rc != GSPLAT_OK) return rc;
    return agree_on_exchange(g);
}

}  // namespace

extern "C" {

int gsplat_group_unique_id(void *id_out) {
    if (!id_out) return GSPLAT_ERR_INVALID_ARGUMENT;
    int rc = load_rccl();
    if (rc != GSPLAT_OK) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return GSPLAT_OK;
}

int gsplat_group_create(gsplat_ctx *ctx, const void *id_bytes, int rank, int world, uint32_t stripe_axis,
                        gsplat_group **out) {
    if (!ctx || !id_bytes || !out || world < 1 || rank < 0 || rank >= world) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int rc = load_rccl();
    if (rc != GSPLAT_OK) return rc;
    gsplat_group *g = new (std::nothrow) gsplat_group();
    if (!g) return GSPLAT_ERR_OUT_OF_MEMORY;
    g->world = world;
    g->members.resize(1);
    g->members[0].ctx = ctx;
    g->members[0].rank = rank;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    if (hipSetDevice(ctx_view(ctx).device) != hipSuccess) { delete g; return GSPLAT_ERR_HIP; }
    ncclResult_t r = g_rccl.CommInitRank(&g->members[0].comm, world, id, rank);
    if (r != ncclSuccess) { delete g; return nccl_fail(r, "ncclCommInitRank"); }
    rc = finish_create(g, stripe_axis);
    if (rc != GSPLAT_OK) { gsplat_group_destroy(g); return rc; }
    *out = g;
    return GSPLAT_OK;
}

int gsplat_group_create_local(gsplat_ctx *const *ctxs, int n, uint32_t stripe_axis, gsplat_group **out) {
    if (!ctxs || !out || n < 1) return GSPLAT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return GSPLAT_ERR_INVALID_ARGUMENT;
    int rc = load_rccl();
    if (rc != GSPLAT_OK) return rc;
    gsplat_group *g = new (std::nothrow) gsplat_group();
    if (!g) return GSPLAT_ERR_OUT_OF_MEMORY;
    g->world = n;
    g->members.resize(n);
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n);
#ifdef GSPLAT_TEST_HOOKS
    // (test builds only — tests/native/fake_rccl.hip moves data inside ONE device, so several members may sit on it; RCCL
    // itself refuses two ranks on one device and so does the shipped library)
    const bool shared_ok = true;
#else
    const bool shared_ok = false;
#endif
    for (int i = 0; i < n; ++i) {
        g->members[i].ctx = ctxs[i];
        g->members[i].rank = i;
        devs[i] = ctx_view(ctxs[i]).device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i] && !shared_ok) { delete g; return set_last_error("two members on one device", GSPLAT_ERR_INVALID_ARGUMENT); }
    }
    ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
    if (r != ncclSuccess) { delete g; return nccl_fail(r, "ncclCommInitAll"); }
    for (int i = 0; i < n; ++i) g->members[i].comm = comms[i];
    rc = finish_create(g, stripe_axis);
    if (rc != GSPLAT_OK) { gsplat_group_destroy(g); return rc; }
    *out = g;
    return GSPLAT_OK;
}

int gsplat_group_set_cuts(gsplat_group *g, const uint32_t *cuts) {
    if (!g || !cuts) return GSPLAT_ERR_INVALID_ARGUMENT;
    const uint32_t extent = g->axis == GSPLAT_STRIPE_COLUMNS ? g->gx : g->gy;
    if (cuts[0] != 0 || cuts[g->world] != extent) return GSPLAT_ERR_OUT_OF_RANGE;
    for (int r = 0; r < g->world; ++r)
        if (cuts[r] > cuts[r + 1]) return GSPLAT_ERR_OUT_OF_RANGE;
    g->cuts.assign(cuts, cuts + g->world + 1);
    return apply_cuts(g);
}

int gsplat_group_exchanges_last_tile(const gsplat_group *g) { return g ? (g->exchange ? 1 : 0) : GSPLAT_ERR_INVALID_ARGUMENT; }

int gsplat_group_render(gsplat_group *g, const gsplat_frame *frame, float *const *outs) {
    if (!g || !frame) return GSPLAT_ERR_INVALID_ARGUMENT;
    const size_t nm = g->members.size();
    std::vector<float4 *> target(nm);
    std::vector<char> begun(nm, 0);
    // A collective that one rank skips blocks every other rank for good.  So from here on a local failure is
    // REMEMBERED, the member still takes part in both exchange steps (with a zero word / whatever its stripe holds) and
    // the first error is returned at the end: the peers get a wrong frame from this rank, not a hang.
    int first_error = GSPLAT_OK;
    char error_text[256] = "";
    auto note = [&](int rc) {
        if (rc != GSPLAT_OK && first_error == GSPLAT_OK) {
            first_error = rc;
            snprintf(error_text, sizeof error_text, "%s", gsplat_last_error());
        }
    };
#define HIP_NOTE(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) note(set_last_error(hipGetErrorString(_e), GSPLAT_ERR_HIP));     \
    } while (0)
#define NCCL_NOTE(expr)                                      \
    do {                                                     \
        ncclResult_t _r = (expr);                            \
        if (_r != ncclSuccess) note(nccl_fail(_r, #expr));   \
    } while (0)
    for (size_t i = 0; i < nm; ++i) {
        const CtxView v = ctx_view(g->members[i].ctx);
        if (v.width != g->width || v.height != g->height)  // (cannot happen while the context refuses gsplat_resize)
            return set_last_error("a member's size differs from the group's", GSPLAT_ERR_INVALID_ARGUMENT);
    }
    // (whether the frame carries the 4-byte exchange was agreed by all ranks at creation: g->exchange, agree_on_exchange)
    const bool exchange = g->exchange;
    // 1. projection, sort on every local member; its own "highest populated tile + 1" lands in its device word
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        target[i] = outs && outs[i] ? reinterpret_cast<float4 *>(outs[i]) : v.image;
        const bool has_tiles = g->cuts[m.rank + 1] > g->cuts[m.rank];
        if (has_tiles) {
            // (blocks that cannot reach the stripe are skipped only if the frame's last tile is exchanged)
            const int rc = ctx_render_begin(m.ctx, frame, m.last_tile, /*stripe_cull=*/exchange);
            note(rc);
            begun[i] = rc == GSPLAT_OK;
        }
        if (!begun[i]) {
            HIP_NOTE(hipSetDevice(v.device));
            HIP_NOTE(hipMemsetAsync(m.last_tile, 0, sizeof(uint32_t), v.stream));
        }
    }
    // 2. the frame's value: 4 bytes, MAX over the members (quirk Q5/Q6, gsplat_boundaries.glsl:39-49)
    if (g->world > 1 && exchange) {
        NCCL_NOTE(g_rccl.GroupStart());
        for (Member &m : g->members) {
            const CtxView v = ctx_view(m.ctx);
            NCCL_NOTE(g_rccl.AllReduce(m.last_tile, m.last_tile, 1, ncclUint32, ncclMax, m.comm, v.stream));
        }
        NCCL_NOTE(g_rccl.GroupEnd());
    }
    // 3. tile ranges + compositor, each stripe at its place in the member's full-frame image
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        if (!begun[i]) continue;
        const int rc = gsplat_render_end(m.ctx, reinterpret_cast<float *>(target[i]), g->width, 0, 0, m.last_tile);
        note(rc);
        ctx_set_last_image(m.ctx, outs && outs[i] ? nullptr : target[i]);
    }
    // 4. all-gather-v of the stripes.  Staged (default): every member packs its stripe — 12 bytes per pixel — into its
    // slot of the stripe-major staging buffer, the peers' slots arrive next to it, one kernel unpacks them all into the
    // row-major image.  Unstaged (GSPLAT_GROUP_PIXELS=rgba, row stripes): contiguous runs of the row-major RGBA image, sent
    // and received in place.
    const size_t fpp = g->rgb ? 3 : 4;  // floats per pixel on the wire
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        HIP_NOTE(hipSetDevice(v.device));
        if (v.timing) HIP_NOTE(hipEventRecord(m.t0, v.stream));
        if (g->staged && g->world > 1) {
            const Rect rc = stripe_rect(g, m.rank);
            const uint32_t px = rc.w * rc.h;
            if (px) {
                float *slot = m.staging + g->stage_off[m.rank] * fpp;
                if (g->rgb)
                    hipLaunchKernelGGL(pack_stripe_kernel<true>, dim3((px + 255u) / 256u), dim3(256), 0, v.stream, target[i],
                                       g->width, rc.x0, rc.y0, rc.w, rc.h, slot);
                else
                    hipLaunchKernelGGL(pack_stripe_kernel<false>, dim3((px + 255u) / 256u), dim3(256), 0, v.stream, target[i],
                                       g->width, rc.x0, rc.y0, rc.w, rc.h, slot);
            }
        }
    }
    if (g->world > 1) {
        // Two forms of the same exchange (GSPLAT_GROUP_GATHER, read at gsplat_group_create):
        //  p2p (default)  every member sends its stripe straight to each peer and receives each peer's: world - 1
        //                 ncclSend + world - 1 ncclRecv per member in one group call.  xGMI is a full mesh of
        //                 point-to-point links (7 per GPU), so every transfer has a link of its own and crosses it once;
        //  broadcast      one grouped ncclBroadcast per stripe (round 3's form): a ring / tree per stripe, whose
        //                 payload passes through the intermediate ranks.
        NCCL_NOTE(g_rccl.GroupStart());
        for (size_t i = 0; i < nm; ++i) {
            Member &m = g->members[i];
            const CtxView v = ctx_view(m.ctx);
            auto slot_of = [&](int r) -> float * {
                return g->staged ? m.staging + g->stage_off[r] * fpp
                                 : reinterpret_cast<float *>(target[i] + (size_t)px_lo(g, r) * g->width);
            };
            for (int peer = 0; peer < g->world; ++peer) {
                const size_t floats = (g->stage_off[peer + 1] - g->stage_off[peer]) * fpp;
                if (floats == 0) continue;
                if (!g->p2p) {
                    NCCL_NOTE(g_rccl.Broadcast(slot_of(peer), slot_of(peer), floats, ncclFloat, peer, m.comm, v.stream));
                } else if (peer != m.rank) {
                    NCCL_NOTE(g_rccl.Recv(slot_of(peer), floats, ncclFloat, peer, m.comm, v.stream));
                }
            }
            if (g->p2p) {
                const size_t floats = (g->stage_off[m.rank + 1] - g->stage_off[m.rank]) * fpp;
                if (floats)
                    for (int peer = 0; peer < g->world; ++peer)
                        if (peer != m.rank) NCCL_NOTE(g_rccl.Send(slot_of(m.rank), floats, ncclFloat, peer, m.comm, v.stream));
            }
        }
        NCCL_NOTE(g_rccl.GroupEnd());
    }
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        HIP_NOTE(hipSetDevice(v.device));
        if (g->staged && g->world > 1) {
            // the peers' stripes into the image: one launch per 8 members (one in all on a node of 8 GPUs)
            for (int c0 = 0; c0 < g->world; c0 += 8) {
                UnpackArgs a;
                memset(&a, 0, sizeof a);
                a.n = g->world - c0 < 8 ? g->world - c0 : 8;
                a.skip = m.rank - c0;  // (outside [0, n): nothing to skip in this chunk)
                for (int k = 0; k < a.n; ++k) {
                    const Rect rc = stripe_rect(g, c0 + k);
                    a.x0[k] = rc.x0; a.y0[k] = rc.y0; a.w[k] = rc.w ? rc.w : 1u; a.h[k] = rc.h;
                    a.first[k] = (uint32_t)(g->stage_off[c0 + k] - g->stage_off[c0]);
                }
                a.first[a.n] = (uint32_t)(g->stage_off[c0 + a.n] - g->stage_off[c0]);
                const uint32_t px = a.first[a.n];
                if (!px) continue;
                const float *chunk = m.staging + g->stage_off[c0] * fpp;
                if (g->rgb)
                    hipLaunchKernelGGL(unpack_stripes_kernel<true>, dim3((px + 255u) / 256u), dim3(256), 0, v.stream, chunk,
                                       g->width, a, target[i]);
                else
                    hipLaunchKernelGGL(unpack_stripes_kernel<false>, dim3((px + 255u) / 256u), dim3(256), 0, v.stream, chunk,
                                       g->width, a, target[i]);
            }
        }
        if (v.timing) {
            HIP_NOTE(hipEventRecord(m.t1, v.stream));
            ctx_record_gather(m.ctx, m.t0, m.t1);
        }
    }
    HIP_NOTE(hipGetLastError());
#undef HIP_NOTE
#undef NCCL_NOTE
    if (first_error != GSPLAT_OK) return set_last_error(error_text, first_error);
    return GSPLAT_OK;
}

// B consecutive frames of every local member through ONE launch sequence (gsplat_render_batch_begin / _end on batch
// contexts), ONE all-reduce of B words and ONE all-gather-v of B stripes per member: the staging buffer holds, member after
// member, that member's B packed stripes one after the other, so a peer's B stripes are one send and one receive.
int gsplat_group_render_batch(gsplat_group *g, const gsplat_frame *frames, uint32_t count) {
    if (!g || !frames || count < 1u) return GSPLAT_ERR_INVALID_ARGUMENT;
    const size_t nm = g->members.size();
    for (size_t i = 0; i < nm; ++i) {
        if (count > ctx_batch_capacity(g->members[i].ctx) || ctx_batch_capacity(g->members[i].ctx) < 2u)
            return set_last_error("gsplat_group_render_batch: a member is not a batch context of that many frames "
                                  "(gsplat_create_batch_view)", GSPLAT_ERR_INVALID_ARGUMENT);
        const CtxView v = ctx_view(g->members[i].ctx);
        if (v.width != g->width || v.height != g->height)
            return set_last_error("a member's size differs from the group's", GSPLAT_ERR_INVALID_ARGUMENT);
    }
    std::vector<char> begun(nm, 0);
    int first_error = GSPLAT_OK;
    char error_text[256] = "";
    auto note = [&](int rc) {
        if (rc != GSPLAT_OK && first_error == GSPLAT_OK) {
            first_error = rc;
            snprintf(error_text, sizeof error_text, "%s", gsplat_last_error());
        }
    };
#define HIP_NOTE(expr)                                                                         \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) note(set_last_error(hipGetErrorString(_e), GSPLAT_ERR_HIP));     \
    } while (0)
#define NCCL_NOTE(expr)                                      \
    do {                                                     \
        ncclResult_t _r = (expr);                            \
        if (_r != ncclSuccess) note(nccl_fail(_r, #expr));   \
    } while (0)
    const bool exchange = g->exchange;
    // 1. projection ... pair sort of the batch on every local member; its frames' own "last tile + 1" words land in m.last_tile[0 .. count)
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        if (g->cuts[m.rank + 1] > g->cuts[m.rank]) {
            const int rc = ctx_batch_begin(m.ctx, frames, count, m.last_tile, /*stripe_cull=*/exchange);
            note(rc);
            begun[i] = rc == GSPLAT_OK;
        }
        if (!begun[i]) {
            HIP_NOTE(hipSetDevice(v.device));
            HIP_NOTE(hipMemsetAsync(m.last_tile, 0, count * sizeof(uint32_t), v.stream));
        }
    }
    // 2. the frames' values: `count` words, MAX over the members, in one all-reduce
    if (g->world > 1 && exchange) {
        NCCL_NOTE(g_rccl.GroupStart());
        for (Member &m : g->members) {
            const CtxView v = ctx_view(m.ctx);
            NCCL_NOTE(g_rccl.AllReduce(m.last_tile, m.last_tile, count, ncclUint32, ncclMax, m.comm, v.stream));
        }
        NCCL_NOTE(g_rccl.GroupEnd());
    }
    // 3. tile ranges + compositor: frame k's stripe at its place in the member's image k
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        if (!begun[i]) continue;
        note(gsplat_render_batch_end(m.ctx, m.last_tile));
    }
    // 4. all-gather-v of the members' `count` stripes each
    const size_t fpp = g->rgb ? 3 : 4;
    const uint32_t image_stride = g->width * g->height;
    std::vector<float4 *> image0(nm);
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        float *p0 = nullptr;
        note(gsplat_batch_image_device_ptr(m.ctx, 0, &p0));
        image0[i] = reinterpret_cast<float4 *>(p0);
        HIP_NOTE(hipSetDevice(v.device));
        if (v.timing) HIP_NOTE(hipEventRecord(m.t0, v.stream));
        if (g->world > 1 && image0[i] != nullptr) {
            const Rect rc = stripe_rect(g, m.rank);
            const uint32_t px = rc.w * rc.h;
            if (px) {
                float *slot = m.staging + (size_t)count * g->stage_off[m.rank] * fpp;
                const dim3 grid((px + 255u) / 256u, count);
                if (g->rgb)
                    hipLaunchKernelGGL(pack_stripe_kernel<true>, grid, dim3(256), 0, v.stream, image0[i], g->width, rc.x0, rc.y0,
                                       rc.w, rc.h, slot, image_stride);
                else
                    hipLaunchKernelGGL(pack_stripe_kernel<false>, grid, dim3(256), 0, v.stream, image0[i], g->width, rc.x0, rc.y0,
                                       rc.w, rc.h, slot, image_stride);
            }
        }
    }
    if (g->world > 1) {
        NCCL_NOTE(g_rccl.GroupStart());
        for (size_t i = 0; i < nm; ++i) {
            Member &m = g->members[i];
            const CtxView v = ctx_view(m.ctx);
            auto slot_of = [&](int r) -> float * { return m.staging + (size_t)count * g->stage_off[r] * fpp; };
            for (int peer = 0; peer < g->world; ++peer) {
                const size_t floats = (g->stage_off[peer + 1] - g->stage_off[peer]) * fpp * count;
                if (floats == 0) continue;
                if (!g->p2p) NCCL_NOTE(g_rccl.Broadcast(slot_of(peer), slot_of(peer), floats, ncclFloat, peer, m.comm, v.stream));
                else if (peer != m.rank) NCCL_NOTE(g_rccl.Recv(slot_of(peer), floats, ncclFloat, peer, m.comm, v.stream));
            }
            if (g->p2p) {
                const size_t floats = (g->stage_off[m.rank + 1] - g->stage_off[m.rank]) * fpp * count;
                if (floats)
                    for (int peer = 0; peer < g->world; ++peer)
                        if (peer != m.rank) NCCL_NOTE(g_rccl.Send(slot_of(m.rank), floats, ncclFloat, peer, m.comm, v.stream));
            }
        }
        NCCL_NOTE(g_rccl.GroupEnd());
    }
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        HIP_NOTE(hipSetDevice(v.device));
        if (g->world > 1 && image0[i] != nullptr) {
            for (int c0 = 0; c0 < g->world; c0 += 8) {
                UnpackArgs a;
                memset(&a, 0, sizeof a);
                a.n = g->world - c0 < 8 ? g->world - c0 : 8;
                a.skip = m.rank - c0;
                for (int k = 0; k < a.n; ++k) {
                    const Rect rc = stripe_rect(g, c0 + k);
                    a.x0[k] = rc.x0; a.y0[k] = rc.y0; a.w[k] = rc.w ? rc.w : 1u; a.h[k] = rc.h;
                    a.first[k] = (uint32_t)(g->stage_off[c0 + k] - g->stage_off[c0]);
                }
                a.first[a.n] = (uint32_t)(g->stage_off[c0 + a.n] - g->stage_off[c0]);
                const uint32_t px = a.first[a.n];
                if (!px) continue;
                const float *chunk = m.staging + (size_t)count * g->stage_off[c0] * fpp;
                const dim3 grid((px + 255u) / 256u, count);
                if (g->rgb)
                    hipLaunchKernelGGL(unpack_stripes_kernel<true>, grid, dim3(256), 0, v.stream, chunk, g->width, a, image0[i],
                                       count, image_stride);
                else
                    hipLaunchKernelGGL(unpack_stripes_kernel<false>, grid, dim3(256), 0, v.stream, chunk, g->width, a, image0[i],
                                       count, image_stride);
            }
        }
        if (v.timing) {
            HIP_NOTE(hipEventRecord(m.t1, v.stream));
            ctx_record_gather(m.ctx, m.t0, m.t1);
        }
    }
    HIP_NOTE(hipGetLastError());
#undef HIP_NOTE
#undef NCCL_NOTE
    if (first_error != GSPLAT_OK) return set_last_error(error_text, first_error);
    return GSPLAT_OK;
}

int gsplat_group_destroy(gsplat_group *g) {
    if (!g) return GSPLAT_OK;
    for (Member &m : g->members) {
        if (m.ctx && m.joined) {
            const CtxView v = ctx_view(m.ctx);
            (void)hipSetDevice(v.device);
            (void)hipStreamSynchronize(v.stream);
            ctx_record_gather(m.ctx, nullptr, nullptr);
            (void)gsplat_set_stripe(m.ctx, GSPLAT_STRIPE_NONE, 0, 0);
            (void)ctx_join_group(m.ctx, nullptr);
        }
        if (m.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m.comm);
        if (m.last_tile) (void)hipFree(m.last_tile);
        if (m.staging) (void)hipFree(m.staging);
        if (m.t0) (void)hipEventDestroy(m.t0);
        if (m.t1) (void)hipEventDestroy(m.t1);
    }
    delete g;
    return GSPLAT_OK;
}

}  // extern "C"
