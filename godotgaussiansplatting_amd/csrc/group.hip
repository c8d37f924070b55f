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

// column stripes are strided in a row-major image: they travel through a contiguous, stripe-major staging buffer
__global__ __launch_bounds__(256) void pack_columns_kernel(const float4 *__restrict__ image, uint32_t pitch, uint32_t x0,
                                                           uint32_t w, uint32_t h, float4 *__restrict__ packed) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < w * h) packed[i] = image[(size_t)(i / w) * pitch + x0 + i % w];
}
__global__ __launch_bounds__(256) void unpack_columns_kernel(const float4 *__restrict__ packed, uint32_t pitch, uint32_t x0,
                                                             uint32_t w, uint32_t h, float4 *__restrict__ image) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < w * h) image[(size_t)(i / w) * pitch + x0 + i % w] = packed[i];
}

struct Member {
    gsplat_ctx *ctx = nullptr;
    int rank = 0;
    ncclComm_t comm = nullptr;
    uint32_t *last_tile = nullptr;   // device word: this member's / the frame's highest populated tile + 1
    float4 *staging = nullptr;       // columns axis: every member's stripe, packed, one after the other
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
    std::vector<size_t> stage_off;    // columns axis: float4 offset of member r's stripe in the staging buffer
    bool p2p = true;                  // the all-gather-v as direct sends / receives (GSPLAT_GROUP_GATHER=broadcast: grouped broadcasts)
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

int apply_cuts(gsplat_group *g) {
    g->stage_off.assign(g->world + 1, 0);
    for (int r = 0; r < g->world; ++r)
        g->stage_off[r + 1] = g->stage_off[r] + (size_t)(px_hi(g, r) - px_lo(g, r)) * g->height;
    for (Member &m : g->members) {
        const int rc = gsplat_set_stripe(m.ctx, g->axis, g->cuts[m.rank], g->cuts[m.rank + 1]);
        if (rc != GSPLAT_OK) return rc;
    }
    return GSPLAT_OK;
}

int finish_create(gsplat_group *g, uint32_t axis) {
    if (axis != GSPLAT_STRIPE_ROWS && axis != GSPLAT_STRIPE_COLUMNS) return GSPLAT_ERR_INVALID_ARGUMENT;
    const CtxView v0 = ctx_view(g->members[0].ctx);
    const char *gm = getenv("GSPLAT_GROUP_GATHER");
    g->p2p = !(gm && !strcmp(gm, "broadcast"));
    g->axis = axis;
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
        if (axis == GSPLAT_STRIPE_COLUMNS)
            HIP_TRY_G(hipMalloc(reinterpret_cast<void **>(&m.staging), (size_t)g->width * g->height * sizeof(float4)));
        HIP_TRY_G(hipEventCreate(&m.t0));
        HIP_TRY_G(hipEventCreate(&m.t1));
    }
    return apply_cuts(g);
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
    // (tests only: GSPLAT_GROUP_SHARED_DEVICE=1 lets several members sit on one device, for a stand-in of RCCL that moves
    // data inside one device — tests/native/fake_rccl.hip; RCCL itself refuses two ranks on one device)
    const char *shared = getenv("GSPLAT_GROUP_SHARED_DEVICE");
    const bool shared_ok = shared && shared[0] == '1';
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

int gsplat_group_render(gsplat_group *g, const gsplat_frame *frame, float *const *outs) {
    if (!g || !frame) return GSPLAT_ERR_INVALID_ARGUMENT;
    const bool columns = g->axis == GSPLAT_STRIPE_COLUMNS;
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
    // Does the frame need the 4-byte exchange?  Only a member that may skip whole blocks of the scene (block culling
    // against its stripe: GSPLAT_FLAG_BLOCK_CULL on a finalized scene) cannot know the frame's highest populated tile
    // by itself; without it every member's own word already is the frame's and the all-reduce — an RCCL launch and a
    // cross-GPU rendezvous in the MIDDLE of every frame — is left out.  The members of a group are created alike (same
    // flags, same scene state) on every rank, so all ranks decide the same way.
    bool exchange = false;
    for (size_t i = 0; i < nm; ++i) {
        const CtxView v = ctx_view(g->members[i].ctx);
        if (v.width != g->width || v.height != g->height)  // (cannot happen while the context refuses gsplat_resize)
            return set_last_error("a member's size differs from the group's", GSPLAT_ERR_INVALID_ARGUMENT);
        exchange = exchange || v.stripe_cull;
    }
    // 1. projection, sort on every local member; its own "highest populated tile + 1" lands in its device word
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        target[i] = outs && outs[i] ? reinterpret_cast<float4 *>(outs[i]) : v.image;
        const bool has_tiles = g->cuts[m.rank + 1] > g->cuts[m.rank];
        if (has_tiles) {
            const int rc = gsplat_render_begin(m.ctx, frame, m.last_tile);
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
    // 4. all-gather-v of the stripes.  Row stripes are contiguous runs of the row-major image (sent and received in
    // place); column stripes go through the packed staging buffer.
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        HIP_NOTE(hipSetDevice(v.device));
        if (v.timing) HIP_NOTE(hipEventRecord(m.t0, v.stream));
        if (columns && g->world > 1) {
            const uint32_t x0 = px_lo(g, m.rank), w = px_hi(g, m.rank) - x0;
            if (w)
                hipLaunchKernelGGL(pack_columns_kernel, dim3((w * g->height + 255u) / 256u), dim3(256), 0, v.stream, target[i],
                                   g->width, x0, w, g->height, m.staging + g->stage_off[m.rank]);
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
            for (int peer = 0; peer < g->world; ++peer) {
                const uint32_t lo = px_lo(g, peer), hi = px_hi(g, peer);
                if (hi <= lo) continue;
                float4 *buf = columns ? m.staging + g->stage_off[peer] : target[i] + (size_t)lo * g->width;
                const size_t floats = (columns ? (size_t)(hi - lo) * g->height : (size_t)(hi - lo) * g->width) * 4;
                if (!g->p2p) {
                    NCCL_NOTE(g_rccl.Broadcast(buf, buf, floats, ncclFloat, peer, m.comm, v.stream));
                } else if (peer != m.rank) {
                    NCCL_NOTE(g_rccl.Recv(buf, floats, ncclFloat, peer, m.comm, v.stream));
                }
            }
            if (g->p2p) {
                const uint32_t lo = px_lo(g, m.rank), hi = px_hi(g, m.rank);
                if (hi > lo) {
                    const float4 *mine = columns ? m.staging + g->stage_off[m.rank] : target[i] + (size_t)lo * g->width;
                    const size_t floats = (columns ? (size_t)(hi - lo) * g->height : (size_t)(hi - lo) * g->width) * 4;
                    for (int peer = 0; peer < g->world; ++peer)
                        if (peer != m.rank) NCCL_NOTE(g_rccl.Send(mine, floats, ncclFloat, peer, m.comm, v.stream));
                }
            }
        }
        NCCL_NOTE(g_rccl.GroupEnd());
    }
    for (size_t i = 0; i < nm; ++i) {
        Member &m = g->members[i];
        const CtxView v = ctx_view(m.ctx);
        HIP_NOTE(hipSetDevice(v.device));
        if (columns && g->world > 1)
            for (int root = 0; root < g->world; ++root) {
                if (root == m.rank) continue;
                const uint32_t x0 = px_lo(g, root), w = px_hi(g, root) - x0;
                if (w)
                    hipLaunchKernelGGL(unpack_columns_kernel, dim3((w * g->height + 255u) / 256u), dim3(256), 0, v.stream,
                                       m.staging + g->stage_off[root], g->width, x0, w, g->height, target[i]);
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
