/*
 * gsplat.h — C ABI of libgsplat_hip.so, the MI355X (gfx950) forward Gaussian-splat rasterizer.
 *
 * Drop-in boundary for the compute-shader pipeline of 2Retr0/GodotGaussianSplatting
 * (util/gaussian_splatting_rasterizer.gd).  The reference exposes a GDScript Resource class, not an
 * FFI; each entry point below names the reference interface it replaces (paths relative to the
 * reference checkout).  A Godot-side GDExtension / C# P/Invoke shim that binds these symbols is shown
 * in INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; every function returns a gsplat_status (0 = OK, negative = error) and never
 *     throws across the boundary;
 *   - one gsplat_ctx = one scene (splat buffer) + one output size on ONE GPU (one process per GPU;
 *     multi-GPU tile-stripe sharding is configured per context with gsplat_set_stripe and the stripes
 *     are gathered by the host with RCCL — see INTEGRATION.md);
 *   - gsplat_render / gsplat_pick / gsplat_resize are not re-entrant per context (call them from one
 *     thread, like Godot's render thread); gsplat_upload_* may run concurrently with each other on
 *     disjoint ranges and with gsplat_render (mirrors ply_file.gd:71 uploading while frames render): chunks go
 *     through a persistent pinned staging ring on a stream of their own, nothing on that path allocates, frees or
 *     waits for the device;
 *   - matrices are column-major float[16] exactly as the reference's 128-byte push constant
 *     (gaussian_splatting_rasterizer.gd:181-193);
 *   - images are RGBA32F, row-major, y down, 16 bytes per pixel (the reference's
 *     R32G32B32A32_SFLOAT storage texture, gaussian_splatting_rasterizer.gd:92).
 */
#ifndef GSPLAT_H
#define GSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSPLAT_VERSION_MAJOR 0
#define GSPLAT_VERSION_MINOR 5

#define GSPLAT_TILE_SIZE 16          /* gaussian_splatting_rasterizer.gd:4, gsplat_render.glsl:8 */
#define GSPLAT_RECORD_FLOATS 60      /* struct Splat, gsplat_projection.glsl:33-40 (240 B) */
#define GSPLAT_PLY_ROW_FLOATS 62     /* INRIA .ply vertex row consumed by ply_file.gd:41-69 */
#define GSPLAT_RASTER_FLOATS 12      /* struct RasterizeData, gsplat_projection.glsl:42-48 (48 B) */
#define GSPLAT_NO_TARGET_TILE 0xFFFFFFFFu /* push constant -1 of gaussian_splatting_rasterizer.gd:158 */

typedef enum gsplat_status {
    GSPLAT_OK = 0,
    GSPLAT_ERR_INVALID_ARGUMENT = -1,
    GSPLAT_ERR_OUT_OF_MEMORY = -2,
    GSPLAT_ERR_HIP = -3,        /* a HIP runtime call failed; see gsplat_last_error() */
    GSPLAT_ERR_NO_DEVICE = -4,
    GSPLAT_ERR_OUT_OF_RANGE = -5,
    GSPLAT_ERR_UNSUPPORTED = -6
} gsplat_status;

/* gsplat_config.flags */
#define GSPLAT_FLAG_TIMING 0x1u        /* record hipEvents around the 4 phases (the reference's
                                          capture_timestamp calls, gaussian_splatting_rasterizer.gd:135-160) */
#define GSPLAT_FLAG_FIX_LAST_TILE 0x2u /* opt out of quirk Q5/Q6 of gsplat_boundaries.glsl:39-49 (off = parity) */
#define GSPLAT_FLAG_KEEP_EMITTED 0x8u  /* keep a copy of the emission-order pairs for GSPLAT_DEBUG_*_EMITTED */
#define GSPLAT_FLAG_KERNEL_TIMING 0x10u /* hipEvents between every launch: per-kernel-class ms in gsplat_stats.ms_kernel */
#define GSPLAT_FLAG_BLOCK_CULL 0x20u   /* after gsplat_finalize_scene: a projection workgroup whose 512 splats lie
                                          outside one frustum plane — or cannot reach this context's stripe — leaves
                                          before reading them.  Invisible in the outputs; a stripe context needs the
                                          gsplat_render_begin/_end form for the stripe part (see there) */
#define GSPLAT_FLAG_FAST_EXP 0x4u      /* compositor uses the hardware exp2 instead of the contract polynomial:
                                          faster, RGBA within 1e-4 except knife-edge pixels */

#define GSPLAT_FLAG_TIES_STORAGE_ORDER 0x40u /* opt-in, contexts on a scene re-laid-out by gsplat_finalize_scene: pairs of EQUAL
                                          key (same tile, same 16-bit depth code) composite in ascending STORAGE slot (the
                                          Morton order) instead of ascending splat id.  The reference reserves key slots
                                          with atomicAdd (gsplat_projection.glsl:196): the order of equal keys is whatever
                                          its waves happened to do, and both orders are members of that family — the frame is
                                          bit for bit the default frame of the same scene uploaded in storage order
                                          (GSPLAT_DEBUG_SLOT_IDS).  What it buys: no tie-repair pass after the sort and
                                          16-bit pair keys (a stripe rank of an 8-GPU frame: -0.03 ms of 0.27).  All contexts
                                          that render parts of one frame (the members of a gsplat_group) must agree on it */

#define GSPLAT_FLAG_READBACK_RGB 0x80u /* gsplat_render_async / gsplat_readback_wait deliver RGB32F — width*height*3 floats, 12
                                          bytes per pixel — instead of RGBA32F: alpha is the constant 1.0 of
                                          gsplat_render.glsl:101, and the 4 bytes it costs per pixel are a quarter of what
                                          crosses PCIe (a 1080p frame: 24.9 MB / 0.45 ms instead of 33.2 MB / 0.60 ms, so the
                                          pipelined rate is bound by the frame, not by the link).  Godot side:
                                          Image.FORMAT_RGBF.  The device image stays RGBA32F */

/* gsplat_config.stripe_axis */
#define GSPLAT_STRIPE_NONE 0u
#define GSPLAT_STRIPE_COLUMNS 1u /* this context owns tile columns [stripe_begin, stripe_end) */
#define GSPLAT_STRIPE_ROWS 2u    /* this context owns tile rows    [stripe_begin, stripe_end) */

typedef struct gsplat_config {
    uint32_t struct_size;       /* = sizeof(gsplat_config) */
    uint32_t max_splats;        /* point_cloud.size (gaussian_splatting_rasterizer.gd:79,83) */
    uint32_t width, height;     /* texture_size (gaussian_splatting_rasterizer.gd:26-29) */
    uint32_t key_budget_factor; /* sort capacity = factor * max_splats; 0 -> 10 (gaussian_splatting_rasterizer.gd:79) */
    int32_t device_id;          /* HIP device ordinal, -1 = current device */
    uint32_t flags;
    uint32_t stripe_axis, stripe_begin, stripe_end;
    int32_t sh_degree;          /* 0..3 = evaluate this many SH bands; -1 = auto (highest band with a non-zero
                                   coefficient among the uploaded splats; zero bands contribute exactly +0) */
    void *stream;               /* hipStream_t to launch on, NULL = a stream owned by the context */
} gsplat_config;

/* Per-frame inputs: the 32-byte uniform block (gsplat_projection.glsl:75-80, written at
 * gaussian_splatting_rasterizer.gd:126), the 128-byte view+projection push constant
 * (gaussian_splatting_rasterizer.gd:181-193) and the render push constant (gsplat_render.glsl:40-43). */
typedef struct gsplat_frame {
    float view[16];
    float proj[16];
    float cam_pos[3];       /* as uploaded by the reference: (-cx, -cy, cz) of the Godot camera origin */
    float model_scale;
    float time;             /* seconds; splat load animation uses time - splat.time */
    float heatmap_factor;   /* float(should_enable_heatmap) */
    uint32_t target_tile;   /* GSPLAT_NO_TARGET_TILE, or the tile whose splat position is picked */
    uint32_t reserved;
} gsplat_frame;

/* kernel classes of one frame, index into gsplat_stats.ms_kernel / launches_kernel */
enum {
    GSPLAT_KERNEL_PROJECT = 0, GSPLAT_KERNEL_SCAN = 1, GSPLAT_KERNEL_EMIT = 2, GSPLAT_KERNEL_SORT_UPSWEEP = 3,
    GSPLAT_KERNEL_SORT_SPINE = 4, GSPLAT_KERNEL_SORT_DOWNSWEEP = 5, GSPLAT_KERNEL_BOUNDARIES = 6,
    GSPLAT_KERNEL_RENDER = 7,
    GSPLAT_KERNEL_SPLAT_SORT = 8, /* the splat-level half of the sort: 2 passes on depth16 over the visible splats */
    GSPLAT_KERNEL_CLASSES = 9
};

/* update_debug_info() of main.gd:93-119 + the roofline inputs of SURVEY.md §8(d).  struct_size is set by the CALLER
 * (= sizeof(gsplat_stats) of the header it was built against): the library fills at most that many bytes, so a binding
 * built against an older header keeps working, and new fields are only ever appended. */
typedef struct gsplat_stats {
    uint32_t struct_size;       /* in: sizeof(gsplat_stats) as the caller knows it (0 is rejected) */
    uint32_t reserved0;
    uint64_t num_splats;        /* N */
    uint64_t num_visible;       /* V: splats that wrote RasterizeData this frame */
    uint64_t num_emitted;       /* D before clamping to the key budget (main.gd:97-100) */
    uint64_t num_sorted;        /* min(D, capacity) */
    uint64_t num_composited;    /* D_c: pairs staged by the compositor before the block early exit */
    uint64_t capacity;
    int32_t overflow;           /* D > capacity ("buffer overflow!", main.gd:100) */
    int32_t sort_passes;
    int32_t sh_degree;          /* bands evaluated */
    int32_t lazy_colors;        /* 1: the compositor evaluated the SH colours of the splats it staged; 0: the
                                   projection pass evaluated them for every visible splat (chosen per frame) */
    float ms_projection, ms_sort, ms_boundaries, ms_render; /* valid with GSPLAT_FLAG_TIMING */
    float ms_total;
    int32_t pair_key_bytes;     /* bytes per key in the pair-level sort of this frame: 2 = the tile id alone (the depth
                                   half of the reference's key is sorted per splat and orders nothing per pair), 4 = the
                                   reference's key (scenes re-laid-out by gsplat_finalize_scene) */
    uint64_t bytes_allocated;   /* device memory behind this context: its own buffers + the scene it renders (main.gd:103) */
    uint64_t scene_bytes;       /* the scene's part of that, shared by every context created with gsplat_create_view */
    uint64_t algorithmic_bytes[4]; /* B_proj, B_sort, B_bounds, B_render (B_render uses D, not D_c) */
    float ms_kernel[GSPLAT_KERNEL_CLASSES];        /* valid with GSPLAT_FLAG_KERNEL_TIMING: summed over the frame's launches */
    uint32_t launches_kernel[GSPLAT_KERNEL_CLASSES];
    uint64_t pairs_round[2];    /* pairs this build emitted and sorted for the frame: [0] alone = num_sorted in a one-round
                                   frame; a two-round frame composites the front of the depth-sorted splats first ([0]) and
                                   emits the rest only where a tile is still unfinished ([1]) — same image, fewer pairs */
    float ms_gather;            /* gsplat_group_render: the exchange of the finished stripes (RCCL), valid with GSPLAT_FLAG_TIMING */
    float ms_readback;          /* gsplat_render_async: the device-to-host copy of the frame, valid with GSPLAT_FLAG_TIMING */
} gsplat_stats;

typedef enum gsplat_debug_buffer {
    GSPLAT_DEBUG_CULLED = 0,        /* RasterizeData[N], 48 B each, indexed by splat id */
    GSPLAT_DEBUG_KEYS_SORTED = 1,   /* u32[num_sorted]: the reference's sorted keys, tile << 16 | depth16.  Taps 1-3 of a
                                       frame that was composited in two rounds (gsplat_stats.pairs_round) are produced by
                                       replaying the frame in one round on this call, from the scene as it is now */
    GSPLAT_DEBUG_VALUES_SORTED = 2, /* u32[num_sorted] */
    GSPLAT_DEBUG_TILE_BOUNDS = 3,   /* uvec2[tiles] */
    GSPLAT_DEBUG_KEYS_EMITTED = 4,  /* u32[num_sorted], emission order of this build: the splats in ascending
                                       (depth16, id), each splat's tiles y-outer/x-inner — the reference's array after
                                       the two depth passes of its sort (needs GSPLAT_FLAG_KEEP_EMITTED) */
    GSPLAT_DEBUG_VALUES_EMITTED = 5,
    GSPLAT_DEBUG_TILE_COUNTS = 6,   /* u32[N] num_tiles_touched per splat (0 = culled) */
    GSPLAT_DEBUG_RECORDS = 7,       /* float[N*60] the scene re-assembled as Splat records */
    GSPLAT_DEBUG_IMAGE = 8,         /* float[W*H*4] the context-owned RGBA32F image */
    GSPLAT_DEBUG_TILE_STAGED = 9,   /* u32[tiles] pairs the compositor staged per tile before its early exit */
    GSPLAT_DEBUG_BLOCK_SUMS = 10,   /* u32[ceil(N/512)][4] per projection workgroup: pairs, visible splats, last tile + 1,
                                       1 if the workgroup was skipped by GSPLAT_FLAG_BLOCK_CULL */
    GSPLAT_DEBUG_TILE_ORDER = 11,   /* u32[tiles of the stripe] the compositor's schedule of the last frame: tile ids,
                                       most expensive first by the staged count of the frame before */
    GSPLAT_DEBUG_SORT_RANK = 12,    /* u32[1]: 1 = the sort's downsweeps rank with returning LDS atomics (the device hands
                                       same-address atomics of a wave out in lane order: checked once per device),
                                       0 = with ballots (GSPLAT_SORT_RANK=ballot, or the check failed) */
    GSPLAT_DEBUG_EMIT_MODE = 13,    /* u32[1]: 1 = the last frame's emission listed its rectangles of more than 512 tiles
                                       for a second launch in which the whole grid shares each of them (frames after one
                                       that met any), 0 = no second launch: every rectangle written by the wave that owns it */
    GSPLAT_DEBUG_SLOT_IDS = 14      /* u32[N]: splat id stored in slot s (the identity until gsplat_finalize_scene) — the
                                       order GSPLAT_FLAG_TIES_STORAGE_ORDER resolves equal keys in */
} gsplat_debug_buffer;

typedef struct gsplat_ctx gsplat_ctx;

/* init_gpu(), gaussian_splatting_rasterizer.gd:65-114: allocate every device buffer for max_splats
 * splats and a width x height output.  The splat buffer starts zeroed.  As in the reference's partially loaded
 * scenes a zero record passes the frustum test while the origin is in view and passes det != 0 (covariance 0 + the 0.3
 * low-pass: det = 0.09, gsplat_projection.glsl:177), but fails the eigenvalue test of :180-181 (0.3 - sqrt(max(0.1, 0))
 * < 0): a not-yet-uploaded splat emits nothing, in the reference and here
 * (tests: test_unloaded_splats_with_the_origin_in_view). */
int gsplat_create(const gsplat_config *config, gsplat_ctx **out_ctx);

/* A second context on the SAME scene (no reference counterpart): its own size / stripe / stream / intermediate
 * buffers / image, but the splat buffer of `scene_owner` — one upload, one copy in HBM however many frames are in
 * flight or stripes are rendered on this GPU (gaussian_splatting_rasterizer.gd:83 holds one scene buffer as well).
 * config->max_splats must be 0 or the owner's; config->device_id is ignored (the owner's device).  Uploads and
 * gsplat_finalize_scene through any of the contexts act on the shared scene; the scene lives until the last
 * context that uses it is destroyed. */
int gsplat_create_view(gsplat_ctx *scene_owner, const gsplat_config *config, gsplat_ctx **out_ctx);

/* cleanup_gpu(), gaussian_splatting_rasterizer.gd:116-120. */
int gsplat_destroy(gsplat_ctx *ctx);

/* device.buffer_update of ply_file.gd:71: `count` 60-float Splat records starting at splat `first`
 * (host or device pointer).  Thread-safe for disjoint ranges. */
int gsplat_upload_splats(gsplat_ctx *ctx, uint32_t first, uint32_t count, const float *records60);

/* PlyFile.load_gaussian_splats, ply_file.gd:28-77: `count` raw INRIA 62-float rows; the swizzle
 * (exp(scale), quaternion -> covariance, sigmoid(opacity), SH re-interleave, ply_file.gd:41-69) runs on
 * the GPU.  load_time is the record's creation_time (ply_file.gd:39). */
int gsplat_upload_ply_rows(gsplat_ctx *ctx, uint32_t first, uint32_t count, const float *rows62, float load_time);

/* Optional, once the scene is loaded (the reference's `loaded` signal, gaussian_splatting_rasterizer.gd:10,114):
 * re-lay the splat storage out along a Morton curve of the positions (acts on the scene: every context created on it
 * with gsplat_create_view sees the new layout).  Purely internal — splat ids, every output and
 * every parity tap are unchanged (equal keys still resolve in ascending splat id — unless the context asked for storage
 * order, GSPLAT_FLAG_TIES_STORAGE_ORDER; exception: WHICH pairs are dropped when the key budget overflows) — but spatially close splats become neighbours in memory, so a tile-stripe shard reads
 * only the cache lines of its own splats (per-rank projection 0.28 -> 0.16 ms at 8 stripes of a 6 M-splat scene) and
 * frustum culling becomes wave-coherent.  Later uploads keep working (they are scattered to the new slots).  Must not
 * run concurrently with gsplat_render. */
int gsplat_finalize_scene(gsplat_ctx *ctx);

/* texture_size setter, gaussian_splatting_rasterizer.gd:26-48 (reallocates tile_bounds and the image). */
int gsplat_resize(gsplat_ctx *ctx, uint32_t width, uint32_t height);

/* Multi-GPU shard (no reference counterpart): restrict this context to a stripe of tiles. */
int gsplat_set_stripe(gsplat_ctx *ctx, uint32_t stripe_axis, uint32_t stripe_begin, uint32_t stripe_end);

/* rasterize(), gaussian_splatting_rasterizer.gd:122-160: projection, key sort, tile ranges, compositor.
 * rgba_out: NULL keeps the frame on the device (gsplat_image_device_ptr); a device pointer is rendered
 * into directly; a host pointer receives a synchronous copy.  width*height*4 floats. */
int gsplat_render(gsplat_ctx *ctx, const gsplat_frame *frame, float *rgba_out);

/* Same as gsplat_render but the frame goes to caller-owned DEVICE memory with an explicit layout: pixel (x, y)
 * of this context's tiles is written to device_out[((y - origin_y) * pitch_px + (x - origin_x)) * 4 .. +3].
 * Used by the multi-GPU host to render a stripe straight into its slot of the all-gather buffer. */
int gsplat_render_to(gsplat_ctx *ctx, const gsplat_frame *frame, float *device_out, uint32_t pitch_px,
                     uint32_t origin_x, uint32_t origin_y);

/* The same frame in two calls, for stripe shards that skip whole blocks of the scene (GSPLAT_FLAG_BLOCK_CULL): quirk
 * Q5/Q6 of gsplat_boundaries.glsl:39-49 needs the frame's highest populated tile, which a rank that no longer projects
 * every splat cannot know by itself.  gsplat_render_begin runs projection, key emission and the sort and stores this
 * context's own "highest populated tile + 1" at last_tile_out_device (4 bytes of device memory, stream-ordered; NULL to
 * skip); the host takes the MAX over the ranks (one 4-byte all-reduce, or nothing on a single GPU) and hands the
 * result to gsplat_render_end, which builds the tile ranges and runs the compositor.  device_out NULL = the
 * context-owned image; frame_last_tile_device NULL = this context's own value (exact for a full-frame context).
 * gsplat_render / gsplat_render_to are begin + end with the context's own value. */
int gsplat_render_begin(gsplat_ctx *ctx, const gsplat_frame *frame, uint32_t *last_tile_out_device);
int gsplat_render_end(gsplat_ctx *ctx, float *device_out, uint32_t pitch_px, uint32_t origin_x, uint32_t origin_y,
                      const uint32_t *frame_last_tile_device);

/* get_splat_position(), gaussian_splatting_rasterizer.gd:162-171: re-runs only the compositor with
 * target_tile = tile_id and reads back {x, y, z, num_tile_splats}; w == 0 means "no splat".
 * Must follow a gsplat_render of the same frame.  The 16-byte result is cleared first.  After a frame
 * that was composited in two rounds the tile's complete sorted list is rebuilt first (a one-round replay of the frame
 * without its compositor): same result, one more frame's worth of sorting on this call. */
int gsplat_pick(gsplat_ctx *ctx, const gsplat_frame *frame, uint32_t tile_id, float out_xyzn[4]);

/* update_debug_info(), main.gd:93-119.  Synchronises with the context's stream. */
int gsplat_get_stats(gsplat_ctx *ctx, gsplat_stats *out);

/* Change the timing flags (GSPLAT_FLAG_TIMING | GSPLAT_FLAG_KERNEL_TIMING) of a live context; other flag bits
 * are fixed at creation and ignored here. */
int gsplat_set_timing(gsplat_ctx *ctx, uint32_t timing_flags);

/* Parity taps for the tests (no reference counterpart).  Copies min(size, available) bytes. */
int gsplat_debug_read(gsplat_ctx *ctx, int which, void *dst, size_t size, size_t *bytes_written);

/* Parity tap of the one transcendental of the projection pass, pow(opacity, 0.2) (gsplat_projection.glsl:190): out_host[i]
 * = the kernels' value for the float whose bit pattern is first_bits + i.  The tests sweep every positive float. */
int gsplat_debug_pow02(gsplat_ctx *ctx, uint32_t first_bits, uint64_t count, float *out_host);

/* Frames for a HOST consumer without stalling the GPU (the drop-in's fallback hand-off when the Godot side cannot
 * import device memory: RenderingDevice.texture_update from a host array).  gsplat_render_async
 * renders the frame into one of two device images and queues its copy into a ring of three pinned host images on a copy
 * stream of its own: the copy of frame k (33 MB at 1080p, ~0.6 ms over PCIe) overlaps the kernels of frame k + 1.
 * *ticket_out identifies the frame; gsplat_readback_wait blocks until that frame is in host memory and returns the
 * pinned image (width*height*4 floats), which stays valid until the third gsplat_render_async after the one that
 * produced it.  Tickets must be waited for in order or skipped; a skipped frame is simply overwritten.
 * With GSPLAT_FLAG_READBACK_RGB the host images are RGB32F (width*height*3 floats).
 * Use ONE context for this per device: two views each running a ring of its own deliver about HALF as many frames as one
 * (measured, tools/d2h_two_contexts.py: 1 595 -> 874 frames/s at 1080p) — the ring already overlaps copy and kernels. */
int gsplat_render_async(gsplat_ctx *ctx, const gsplat_frame *frame, uint64_t *ticket_out);
int gsplat_readback_wait(gsplat_ctx *ctx, uint64_t ticket, const float **host_rgba_out);

/* The drop-in's primary hand-off: render straight into memory the HOST's graphics API owns.  `fd` is an opaque POSIX
 * file descriptor exported for the Vulkan image / buffer behind the Texture2DRD of
 * gaussian_splatting_rasterizer.gd:92,101 (vkGetMemoryFdKHR, VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT; on amdgpu
 * that is a dma-buf) — or a dma-buf of any other device allocation; size_bytes = the allocation's size, offset_bytes =
 * where the linear RGBA32F image (row-major, pitch = width) starts in it.  The memory is imported with
 * hipImportExternalMemory and becomes the target of every later gsplat_render(ctx, frame, NULL) /
 * gsplat_image_device_ptr; the library takes ownership of fd.  gsplat_bind_external_image(ctx, -1, 0, 0) unbinds.
 * Synchronisation with the consumer is the caller's (gsplat_synchronize, or a shared semaphore on ctx's stream). */
int gsplat_bind_external_image(gsplat_ctx *ctx, int fd, uint64_t size_bytes, uint64_t offset_bytes);
/* Counterpart for tests and for HIP/Vulkan hosts that allocate on the HIP side: a dma-buf file descriptor of the
 * context-owned image (hipMemGetHandleForAddressRange), which the other API imports.  The caller closes *fd_out. */
int gsplat_export_image_fd(gsplat_ctx *ctx, int *fd_out, uint64_t *size_bytes_out);

/* ---- Multi-GPU: one frame sharded by tile stripes over the GPUs of a node (no reference counterpart) ----
 * The scene is replicated (every member context holds all splats); member r owns a contiguous stripe of tile rows (or
 * columns) and clamps every splat's tile rectangle to it, so the per-tile pair lists — hence the pixels — are exactly
 * the single-GPU frame's.  Two exchange steps per frame, both inside the library, on the members' own streams (RCCL over
 * xGMI): a 4-byte all-reduce(MAX) of "highest populated tile + 1" between gsplat_render_begin and gsplat_render_end
 * (quirk Q5/Q6 of gsplat_boundaries.glsl:39-49 belongs to the FRAME's last tile) — carried by a group's frames only when a
 * member may skip whole blocks of the scene against its stripe (GSPLAT_FLAG_BLOCK_CULL on a finalized scene: otherwise every
 * member's own value already is the frame's); whether it is, is AGREED ONCE by all ranks inside gsplat_group_create (MAX over
 * the ranks; gsplat_group_exchanges_last_tile) and never decided again per frame from a rank's own state: a member whose
 * state differs later renders without the stripe part of the culling instead of leaving its peers in a collective —
 * and an all-gather-v of the finished stripes that leaves the complete RGBA32F frame in every member's image: every member
 * packs the three colour channels of its stripe (12 bytes per pixel travel: alpha is the constant 1.0 of
 * gsplat_render.glsl:101 and is rebuilt on arrival; GSPLAT_GROUP_PIXELS=rgba sends 16), sends them straight to each peer and
 * receives each peer's (grouped ncclSend / ncclRecv: xGMI is a full mesh of point-to-point links, every transfer crosses
 * one link once; unequal stripes need no padding; GSPLAT_GROUP_GATHER=broadcast selects one grouped ncclBroadcast per
 * stripe instead), and one kernel unpacks all of them.  librccl is loaded on first use (GSPLAT_RCCL_LIB overrides the name);
 * a single-GPU program never touches it.
 * While a context is a member of a group it refuses gsplat_resize and gsplat_destroy (the group caches its size and
 * pointer) and cannot join a second group: gsplat_group_destroy first.  If a member's frame fails locally,
 * gsplat_group_render still takes part in both exchange steps — a collective one rank skips blocks every other rank
 * for good — and returns the first error afterwards; the peers then hold a wrong stripe of that member, not a hang.
 *   one process per GPU:  rank 0 calls gsplat_group_unique_id and hands the 128 bytes to the other ranks (any channel);
 *                         every rank calls gsplat_group_create(ctx, id, rank, world, axis, &g) — collective;
 *   one process, n GPUs:  gsplat_group_create_local(ctxs, n, axis, &g) with one context per device (the host
 *                         north_star names — Godot's single render thread — shards without spawning processes).
 *                         (Two members on one device are refused, as RCCL would; only a TEST build of the library —
 *                         group.hip compiled with -DGSPLAT_TEST_HOOKS — lifts that, for the suite's stand-in of RCCL.)
 * gsplat_group_render renders the frame on every LOCAL member and returns when the work is queued; afterwards (stream
 * order / gsplat_synchronize) each member's image (gsplat_image_device_ptr, or outs[i] if given: device pointers on the
 * members' devices, width*height*4 floats) holds the whole frame.  gsplat_stats.ms_gather of a member times the exchange.
 * gsplat_group_set_cuts moves the stripe boundaries (world + 1 ascending tile indices from 0 to the grid's extent
 * along the axis, identical on every rank) — e.g. to balance the members by last frame's pairs per tile row. */
typedef struct gsplat_group gsplat_group;
#define GSPLAT_GROUP_ID_BYTES 128
int gsplat_group_unique_id(void *id_out /* GSPLAT_GROUP_ID_BYTES */);
int gsplat_group_create(gsplat_ctx *ctx, const void *id, int rank, int world, uint32_t stripe_axis, gsplat_group **out);
int gsplat_group_create_local(gsplat_ctx *const *ctxs, int n, uint32_t stripe_axis, gsplat_group **out);
int gsplat_group_set_cuts(gsplat_group *group, const uint32_t *cuts /* world + 1 */);
int gsplat_group_render(gsplat_group *group, const gsplat_frame *frame, float *const *outs /* per local member, or NULL */);
/* 1 / 0: does every frame of this group carry the 4-byte last-tile all-reduce?  Agreed by ALL ranks inside
 * gsplat_group_create (MAX over the ranks of "my member may skip blocks against its stripe"), fixed for the group's life. */
int gsplat_group_exchanges_last_tile(const gsplat_group *group);
/* `count` consecutive frames of every LOCAL member — batch contexts (gsplat_create_batch_view) of at least that many frames —
 * through one launch sequence each, ONE all-reduce of `count` words and ONE all-gather-v in which a member's `count` stripes
 * travel as one message per peer; afterwards image k of every member (gsplat_batch_image_device_ptr) holds frame k whole. */
int gsplat_group_render_batch(gsplat_group *group, const gsplat_frame *frames, uint32_t count);
int gsplat_group_destroy(gsplat_group *group);

/* ---- Batched frames (no reference counterpart; SURVEY.md §8e "what does not shrink with G") ----
 * B consecutive frames of ONE context — B cameras, one scene, one stripe — through ONE launch sequence.  What a stripe
 * rank of a multi-GPU frame spends its time on is per LAUNCH, not per frame: ten of its fourteen launches take 5 - 15 us
 * whatever their input, and its compositor is bound by the serial chain of its heaviest tile while most of the chip idles.
 * A batch is rendered as one frame of a virtual image that stacks the B stripes vertically (B x the splats, B x the tile
 * rows): the latency-bound launches serve B frames and the compositor has B x the tiles to fill the chip with.  Every frame
 * of a batch is bit for bit the frame gsplat_render would have produced (tests: test_batched_frames_*).
 *   gsplat_create_batch_view   a context on scene_owner's scene whose intermediate buffers hold `batch` frames (1 .. 4):
 *                              per-splat arrays, pair buffers (key budget = batch x the frame's: only the batch's TOTAL can
 *                              overflow) and `batch` images.  Needs 16-bit pair keys: a scene in upload order, or
 *                              GSPLAT_FLAG_TIES_STORAGE_ORDER on a re-laid-out one (GSPLAT_ERR_UNSUPPORTED otherwise), and
 *                              batch x stripe rows x tile columns <= 65 536.  The context still renders plain frames.
 *   gsplat_render_batch        count <= batch frames; frame k lands in image k (gsplat_batch_image_device_ptr; the stripe's
 *                              pixels at their place in a full-frame image, like gsplat_render on a stripe context).
 *   gsplat_render_batch_begin / _end   the same in two calls, as gsplat_render_begin / _end: `count` words of "highest
 *                              populated tile + 1" out, the frames' (MAX over the ranks, word by word) in.
 * One round per frame (no two-round frames), no pick and no sort / record taps after a batch (render the frame alone). */
#define GSPLAT_MAX_BATCH 4
int gsplat_create_batch_view(gsplat_ctx *scene_owner, const gsplat_config *config, uint32_t batch, gsplat_ctx **out_ctx);
int gsplat_render_batch(gsplat_ctx *ctx, const gsplat_frame *frames, uint32_t count);
int gsplat_render_batch_begin(gsplat_ctx *ctx, const gsplat_frame *frames, uint32_t count, uint32_t *last_tiles_out_device);
int gsplat_render_batch_end(gsplat_ctx *ctx, const uint32_t *frame_last_tiles_device);
int gsplat_batch_image_device_ptr(gsplat_ctx *ctx, uint32_t index, float **out_ptr);

/* Device pointer of the context-owned RGBA32F image (the Texture2DRD of gaussian_splatting_rasterizer.gd:92). */
int gsplat_image_device_ptr(gsplat_ctx *ctx, float **out_ptr);

/* Wait for everything queued on the context's stream. */
int gsplat_synchronize(gsplat_ctx *ctx);

/* update_camera_matrices(), gaussian_splatting_rasterizer.gd:175-195.  camera_xform: the camera-to-world
 * transform (basis columns X,Y,Z then origin, 12 floats); basis_override: 9 floats (columns) or NULL.
 * Perspective = Godot 4.3 Camera3D.get_camera_projection(): fovy in degrees, keep-height aspect.
 * out32 = view[16] | proj[16]; out_cam_pos = the 3 floats of the uniform block (may be NULL). */
int gsplat_make_view_proj(const float camera_xform[12], const float basis_override[9], float fovy_degrees,
                          float aspect, float z_near, float z_far, float out32[32], float out_cam_pos[3]);

const char *gsplat_status_string(int status);
const char *gsplat_last_error(void); /* thread-local detail of the last GSPLAT_ERR_HIP */
uint32_t gsplat_version(void);       /* (major << 16) | minor */

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_H */
