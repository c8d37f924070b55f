// Projection + key emission for gfx950 — replaces resources/shaders/compute/gsplat_projection.glsl.
//
// One lane per splat, 512-lane workgroups (8 wave64).  The scene is SoA (SceneSoA) so every load
// instruction of a wave is one contiguous 1 KiB run; culled splats touch 16 B.  The reference reserves
// key slots with a global atomicAdd (gsplat_projection.glsl:196), which makes the order of equal keys
// non-deterministic; the contract's member is "ascending splat id, y-outer/x-inner" (DESIGN.md §3).
//
// Every pair of a splat carries the splat's 16-bit depth code, so the two low radix passes of the reference's
// sort are passes over splats (sort.hip).  The frame therefore runs
//   project_kernel       cull, project, RasterizeData, per-splat hand-off {depth16 | origin tile, rectangle size}
//                        and the workgroup's histogram of the low depth byte (pass 0 of the splat sort)
//   launch_sort_splats   the visible splats in (depth16, id) order                                  (sort.hip)
//   emit_sums_kernel     pairs per 512-splat block of that order
//   scan_blocks_kernel   scan of the workgroup totals, D, overflow, frame counters, clears tile_bounds
//   emit_kernel          (tile<<16 | depth16, id) pairs, y-outer/x-inner (gsplat_projection.glsl:218-226), written
//                        in (depth16, id) order = the reference's array after its second sort pass; with 16-bit keys
//                        the key is the tile id INSIDE the context's stripe (TileMap: same order, fewer bits to sort)
// The SH colour (get_color, :94-121) is evaluated here in "eager" frames, for every visible splat; in "lazy" frames
// the compositor evaluates it for the splats it stages (sh_eval.h: one shared expression; api.hip chooses per frame).
//
// Arithmetic follows the contract in DESIGN.md §3 (compile with -ffp-contract=off): IEEE binary32,
// left-to-right sums, correctly rounded / and sqrt, pow(x,0.2) as a binary64 fifth root.
#include "gsplat_internal.h"
#include "project_math.h"
#include "sh_eval.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace gsplat {

namespace {

// pow(x, 0.2), gsplat_projection.glsl:190.  The contract (the checker restates it) is the
// binary64 fifth root by 5 Newton steps r <- (4r + x / r^4) / 5 from a bit-level guess, rounded once to binary32 — ten
// binary64 divisions per splat, which is most of this kernel's VALU time once a lazy frame no longer writes records.
// The same binary32 value for less: a division-free Newton iteration on the INVERSE fifth root, y <- y (6 - x y^5) / 5
// (error e -> 3 e^2), two steps from the hardware's exp2(-log2(x) / 5) (1e-6 -> 3e-12 -> 3e-23), r = x y^4.  That r and
// the contract's are both within a few 2^-53 of the real root, so they round to the same binary32 number unless the root
// lies within ~2^-50 of a rounding boundary: a result whose low 29 mantissa bits are within 2^15 of the boundary pattern
// (one lane in 8 000), tiny, huge or non-finite inputs take the contract's own loop.  Verified bit for bit against the
// oracle on ALL 2 139 095 039 positive finite floats (tests/test_gpu_parity.py::test_pow02_exhaustive).
__device__ __forceinline__ float pow02_contract(float xf) {
    const double x = (double)xf;
    long long i = __double_as_longlong(x);
    const long long B = 0x3FF0000000000000LL;
    i = i / 5 + (B - B / 5);
    double r = __longlong_as_double(i);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const double r2 = r * r;
        const double r4 = r2 * r2;
        r = (4.0 * r + x / r4) / 5.0;
    }
    return (float)r;
}

__device__ __forceinline__ float pow02(float xf) {
    if (!(xf > 0.0f)) return 0.0f;
    bool slow = !(xf >= 0x1p-100f && xf <= 0x1p+100f);  // (hardware log2 / exp2 flush denormals; inf)
    float result = 0.0f;
    if (!slow) {
        const double x = (double)xf;
        double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * -0.2f);  // ~ x^(-1/5)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double y2 = y * y;
            const double y5 = (y2 * y2) * y;
            y = (y * __builtin_fma(-x, y5, 6.0)) * 0.2;
        }
        const double y2 = y * y;
        const double r = x * (y2 * y2);
        const unsigned long long low = (unsigned long long)__double_as_longlong(r) & 0x1FFFFFFFull;  // below binary32
        slow = low - (0x10000000ull - 0x8000ull) <= 0x10000ull;  // within 2^15 of the round-to-nearest boundary
        result = (float)r;
    }
    if (slow) result = pow02_contract(xf);
    return result;
}

// parity tap: pow02 of the floats whose bit patterns are first_bits, first_bits + 1, ...
__global__ __launch_bounds__(256) void pow02_bits_kernel(uint32_t first_bits, uint64_t count, float *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256u)
        out[i] = pow02(__uint_as_float(first_bits + (uint32_t)i));
}

// Everything gsplat_projection.glsl:150-206 does for one splat: cull, project, colour, write RasterizeData.
// Returns num_tiles_touched (0 = the splat emits nothing); key_out = depth16 | (tile id of the rectangle's first
// tile) << 16, dims_out = w | h << 16 of the rectangle clamped to the stripe, last_plus1 = last tile of the unclamped
// rectangle + 1.  EAGER >= 0: the colour is evaluated here with bands 0..EAGER; -1: left to the compositor.
// RECORD: produce the RasterizeData record (a lazy frame does not: its compositor recomputes the geometry half from the
// scene for the splats it stages, project_math.h, and evaluates their colours — the record is only built for the tap).
template <int EAGER, bool RECORD>
__device__ __forceinline__ uint32_t project_splat(const SceneSoA &scene, uint32_t n, const FrameParams &fp, uint32_t id,
                                                  float4 (&record)[3], uint32_t &key_out, uint32_t &dims_out,
                                                  uint32_t &last_plus1_out) {
    uint32_t count = 0, last_plus1 = 0;
    uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0, depth16 = 0;
    float ipx = 0, ipy = 0;
    ClipPos cp{};
    Footprint ft{};
    ft.det = 1.0f;

    if (id < n) {
        const float4 pt = scene.pos_time[id];
        cp = splat_clip(fp, pt);
        if (!splat_outside_frustum(cp)) {  // :160-166 frustum culling
            const float4 A = scene.cov_a[id];
            const float4 Bc = scene.cov_b[id];
            ft = splat_footprint(fp, cp, pt.w, A, Bc);  // :169-174, :124-142
            // :177-182
            const float mid = 0.5f * (ft.ca + ft.cc);
            const float disc = sqrtf(fmaxf(0.1f, mid * mid - ft.det));
            const float l1 = mid + disc, l2 = mid - disc;
            if (ft.det != 0.0f && !(l1 < 0.0f) && !(l2 < 0.0f)) {
                splat_image_pos(fp, cp, ft.tf, ipx, ipy);  // :184-185
                const float nz = cp.cz / cp.cw;
                // :190-194, get_rect :144-148
                const float radius = (pow02(ft.opacity) * 2.5f) * sqrtf(fmaxf(l1, l2));
                const float gxf = (float)fp.gx, gyf = (float)fp.gy;
                x0 = (uint32_t)(int32_t)clampf((ipx - radius) / 16.0f, 0.0f, gxf);
                y0 = (uint32_t)(int32_t)clampf((ipy - radius) / 16.0f, 0.0f, gyf);
                x1 = (uint32_t)(int32_t)clampf(ceilf((ipx + radius) / 16.0f), 0.0f, gxf);
                y1 = (uint32_t)(int32_t)clampf(ceilf((ipy + radius) / 16.0f), 0.0f, gyf);
                // last tile of the unclamped rectangle: every shard sees the whole frame's highest populated
                // tile, the only one quirk Q5/Q6 may hit
                if (x1 > x0 && y1 > y0) last_plus1 = (y1 - 1) * fp.gx + (x1 - 1) + 1;
                x0 = max(x0, fp.sx0); y0 = max(y0, fp.sy0);
                x1 = min(x1, fp.sx1); y1 = min(y1, fp.sy1);
                if (x1 > x0 && y1 > y0) {
                    count = (x1 - x0) * (y1 - y0);
                    depth16 = (uint32_t)(((nz * nz) * nz) * 65535.0f) & 0xFFFFu;  // :218
                }
            }
        }
    }

    if (RECORD && count) {
        // :202-206 RasterizeData.  The colour (:198-201, get_color): eager frames evaluate it here — band-0 scenes from the
        // streamed band-0 plane (16 B per splat), scenes with higher bands from the splat's 192-byte coefficient block;
        // lazy frames leave it to the compositor, which only evaluates the splats it stages (at 6 M splats / deg 3 half
        // of the visible splats never are, and the coefficients would be 60 % of this kernel's traffic).
        float rgb[3] = {0.0f, 0.0f, 0.0f};
        if (EAGER == 0) {
            const float4 dc = scene.sh_dc[id];
            const float c[3] = {dc.x, dc.y, dc.z};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) rgb[ch] = sh_channel<0>(&c[ch], 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
        } else if (EAGER > 0) {
            float x, y, z;
            sh_direction(cp.px, cp.py, cp.pz, fp.cam, x, y, z);
            sh_rgb_wide<(EAGER > 0 ? EAGER : 1)>(scene.sh_block + (size_t)id * SH_BLOCK_F4, x, y, z, rgb);
        }
        splat_raster_geometry(cp, ft, ipx, ipy, record[0], record[1]);  // image_pos, pos_xy | conic, pos_z
        record[2] = make_float4(rgb[0], rgb[1], rgb[2], ft.opacity);    // color, opacity
    }
    key_out = depth16 | ((y0 * fp.gx + x0) << 16);
    dims_out = count ? ((x1 - x0) | ((y1 - y0) << 16)) : 0u;
    last_plus1_out = last_plus1;
    return count;
}

// wave64 inclusive scan (shuffle-up ladder)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Workgroup-level culling of a spatially ordered scene (gsplat_finalize_scene + GSPLAT_FLAG_BLOCK_CULL).
// A projection workgroup owns 512 consecutive storage slots = a compact region after the Morton re-layout;
// block_bounds_kernel records its axis-aligned box, the largest |covariance|_F, the largest opacity factor and the
// latest load time.  block_outside() decides from those 48 bytes whether NO splat of the workgroup can emit a pair,
// in two steps that are both conservative against the f32 evaluation in project_splat:
//  1. all 8 box corners are outside the same plane of the reference's frustum test (gsplat_projection.glsl:160-166;
//     each test is affine in the position, so the box is outside if its corners are) — valid always;
//  2. (cull_mode 2) every splat is in its steady state (time - load_time > 1.35 s: tf = tfl = 1), all corners are in
//     front of the camera, and the screen interval of the box, widened by a bound R of the tile-rectangle radius,
//     misses the context's stripe.  R: radius = pow(opacity,0.2) * 2.5 * sqrt(l1) (:181-190) with
//     l1 <= lambda_max(T S T^t) + 0.3 + sqrt(0.1), lambda_max(T S T^t) <= |J|_F^2 |W|_2^2 rho(S),
//     |J|_F^2 <= (fx^2 + fy^2 (1 + 1.69/P00^2 + 1.69/P11^2)) / z_min^2 (the clamp of :132 bounds m), rho(S) <= |S|_F *
//     model_scale^2; +0.1 % and +1 px cover the f32 rounding of the real evaluation.
// A culled workgroup contributes no pairs, no visible splats and no "last tile" — for step 2 in a stripe context
// the frame's last tile therefore has to come from the host (gsplat_render_end).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool block_outside(const FrameParams &fp, const float4 b0, const float4 b1, const float4 b2) {
    const float *V = fp.V, *P = fp.P;
    const float ms = fp.model_scale;
    bool out_l = true, out_r = true, out_b = true, out_t = true, out_n = true, out_f = true;
    float vz_max = -INFINITY, cw_min = INFINITY, noise = 0.0f;
    float nx_min = INFINITY, nx_max = -INFINITY, ny_min = INFINITY, ny_max = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float px = ((k & 1) ? b1.x : b0.x) * ms, py = ((k & 2) ? b1.y : b0.y) * ms, pz = ((k & 4) ? b1.z : b0.z) * ms;
        const float vx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
        const float vy = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
        const float vz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
        const float vw = ((V[3] * px + V[7] * py) + V[11] * pz) + V[15];
        const float cx = ((P[0] * vx + P[4] * vy) + P[8] * vz) + P[12] * vw;
        const float cy = ((P[1] * vx + P[5] * vy) + P[9] * vz) + P[13] * vw;
        const float cz = ((P[2] * vx + P[6] * vy) + P[10] * vz) + P[14] * vw;
        const float cw = ((P[3] * vx + P[7] * vy) + P[11] * vz) + P[15] * vw;
        // the same sums with every term's magnitude: (a few) 2^-24 of these bound the rounding of project_splat's own
        // evaluation at any point of the box, cancellation included; 1e-5 of them is the margin
        const float ax = fabsf(px), ay = fabsf(py), az = fabsf(pz);
        const float avx = ((fabsf(V[0]) * ax + fabsf(V[4]) * ay) + fabsf(V[8]) * az) + fabsf(V[12]);
        const float avy = ((fabsf(V[1]) * ax + fabsf(V[5]) * ay) + fabsf(V[9]) * az) + fabsf(V[13]);
        const float avz = ((fabsf(V[2]) * ax + fabsf(V[6]) * ay) + fabsf(V[10]) * az) + fabsf(V[14]);
        const float avw = ((fabsf(V[3]) * ax + fabsf(V[7]) * ay) + fabsf(V[11]) * az) + fabsf(V[15]);
        const float acx = ((fabsf(P[0]) * avx + fabsf(P[4]) * avy) + fabsf(P[8]) * avz) + fabsf(P[12]) * avw;
        const float acy = ((fabsf(P[1]) * avx + fabsf(P[5]) * avy) + fabsf(P[9]) * avz) + fabsf(P[13]) * avw;
        const float acz = ((fabsf(P[2]) * avx + fabsf(P[6]) * avy) + fabsf(P[10]) * avz) + fabsf(P[14]) * avw;
        const float acw = ((fabsf(P[3]) * avx + fabsf(P[7]) * avy) + fabsf(P[11]) * avz) + fabsf(P[15]) * avw;
        const float vb = cw * 1.2f;
        const float mx = 1e-5f * (acx + 1.2f * acw), my = 1e-5f * (acy + 1.2f * acw), mz = 1e-5f * (acz + acw);
        out_l = out_l && (cx < -vb - mx);
        out_r = out_r && (cx > vb + mx);
        out_b = out_b && (cy < -vb - my);
        out_t = out_t && (cy > vb + my);
        out_n = out_n && (cz < -mz);
        out_f = out_f && (cz > cw + mz);
        vz_max = fmaxf(vz_max, vz + 1e-5f * avz);
        cw_min = fminf(cw_min, cw - 1e-5f * acw);
        noise = fmaxf(noise, fmaxf(acx, acy) + acw);
        const float nx = cx / cw, ny = cy / cw;
        nx_min = fminf(nx_min, nx); nx_max = fmaxf(nx_max, nx);
        ny_min = fminf(ny_min, ny); ny_max = fmaxf(ny_max, ny);
    }
    if (out_l || out_r || out_b || out_t || out_n || out_f) return true;
    if (fp.cull_mode < 2u) return false;
    if (!(fp.time - b2.x > 1.36f)) return false;   // load animation may still move or inflate a splat
    if (!(cw_min > 0.0f && vz_max < 0.0f)) return false;
    const float inv = 1.0f / (-vz_max);
    const float fx = (fp.Wf * 0.5f) * fabsf(P[0]) * inv, fy = (fp.Hf * 0.5f) * fabsf(P[5]) * inv;
    const float mxb = 1.3f / fabsf(P[0]), myb = 1.3f / fabsf(P[5]);
    const float j2 = fx * fx + (fy * fy) * ((1.0f + mxb * mxb) + myb * myb);
    const float lam = ((j2 * fp.view_norm2) * b0.w) * (ms * ms) + 0.62f;
    // + rounding of the centre's screen position (ndc error <= ~2^-22 * noise / cw, see above)
    const float R = ((2.5f * b1.w) * sqrtf(lam)) * 1.001f + 1.0f + (1e-5f * fmaxf(fp.Wf, fp.Hf)) * (noise / cw_min);
    const float x_lo = ((nx_min + 1.0f) * 0.5f) * fp.Wm1 - R, x_hi = ((nx_max + 1.0f) * 0.5f) * fp.Wm1 + R;
    const float y_lo = ((ny_min + 1.0f) * 0.5f) * fp.Hm1 - R, y_hi = ((ny_max + 1.0f) * 0.5f) * fp.Hm1 + R;
    // NaN anywhere makes every comparison false: the workgroup is kept
    return x_hi < 16.0f * (float)fp.sx0 || x_lo > 16.0f * (float)fp.sx1 || y_hi < 16.0f * (float)fp.sy0 ||
           y_lo > 16.0f * (float)fp.sy1;
}

__global__ __launch_bounds__(PROJ_BLOCK) void block_bounds_kernel(SceneSoA scene, uint32_t n,
                                                                  float4 *__restrict__ block_bounds) {
    __shared__ float red[PROJ_BLOCK / 64][9];
    const uint32_t id = blockIdx.x * PROJ_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[9] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0f, 1.0f, -INFINITY};
    if (id < n) {
        const float4 pt = scene.pos_time[id], A = scene.cov_a[id], B = scene.cov_b[id];
        const float diag = (A.x * A.x + A.w * A.w) + B.y * B.y, off = (A.y * A.y + A.z * A.z) + B.x * B.x;
        float F = sqrtf(diag + 2.0f * off) * 1.00001f;
        const float op = B.z;
        // anything that is not an ordinary record (NaN/inf, negative opacity) switches culling off for the workgroup
        const bool ok = isfinite(pt.x) && isfinite(pt.y) && isfinite(pt.z) && isfinite(pt.w) && isfinite(F) &&
                        op >= 0.0f && isfinite(op);
        if (!ok) F = INFINITY;
        v[0] = v[3] = pt.x; v[1] = v[4] = pt.y; v[2] = v[5] = pt.z;
        v[6] = F;
        v[7] = op > 1.0f ? op : 1.0f;  // >= max(1, op)^0.2
        v[8] = pt.w;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float o = __shfl_xor(v[k], d, 64);
            v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
        }
    }
    // fminf/fmaxf drop NaNs: positions were checked above (F = inf) so nothing is lost
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 9; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < PROJ_BLOCK / 64; ++w)
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = k < 3 ? fminf(v[k], red[w][k]) : fmaxf(v[k], red[w][k]);
        block_bounds[3 * blockIdx.x + 0] = make_float4(v[0], v[1], v[2], v[6]);
        block_bounds[3 * blockIdx.x + 1] = make_float4(v[3], v[4], v[5], v[7]);
        block_bounds[3 * blockIdx.x + 2] = make_float4(v[8], 0.0f, 0.0f, 0.0f);
    }
}

// one thread per projection workgroup, once per frame (12 k threads at 6 M splats, a few us): evaluating the 8 corners
// inside project_kernel itself made every one of its 8 waves pay ~1000 instructions and turned the HBM-bound kernel
// VALU-bound (0.38 -> 0.61 ms)
__global__ __launch_bounds__(256) void block_cull_kernel(FrameParams fp, const float4 *__restrict__ block_bounds,
                                                         uint32_t num_blocks, uint32_t *__restrict__ block_skip) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= num_blocks) return;
    block_skip[b] = block_outside(fp, block_bounds[3 * b], block_bounds[3 * b + 1], block_bounds[3 * b + 2]) ? 1u : 0u;
}
// The workgroups block_cull_kernel did NOT skip, as compact ascending lists: blocks at live[LIVE_HEADER ..], the partitions of
// splat-sort pass 0 (bpp blocks each) that hold at least one at live[LIVE_HEADER + num_blocks ..]; live[0] / live[1] = how many.
// Why: project_kernel and pass 0 hand XCD x the CONTIGUOUS eighth x of the blocks / partitions (their scattered writes meet
// in one L2), and a stripe rank skips two thirds of its blocks — the live ones cluster along the Morton curve, so one XCD got
// up to 3.4 x the mean (tools/live_blocks_probe.py: c3 rows 0:17 -> [366, 184, 0, 252, 740, 418, 10, 10] live blocks per
// XCD) and the launch lasted as long as that XCD.  Dealing the eighths of the LIST keeps the locality and evens the load.
// One workgroup: thread t counts its run of ceil(n / 1024) consecutive entries, one scan, then writes them.  Also gives
// every skipped block its (0, 0, 0, 1) record — the scan reads all blocks' records, and no projection workgroup visits
// a skipped block any more.
constexpr uint32_t LIVE_HEADER = 8;
__global__ __launch_bounds__(1024) void live_lists_kernel(const uint32_t *__restrict__ skip, uint32_t num_blocks, uint32_t bpp,
                                                          uint32_t *__restrict__ live, uint4 *__restrict__ block_sums) {
    __shared__ uint32_t wave_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *out = live + LIVE_HEADER;
#pragma unroll 1
    for (int level = 0; level < 2; ++level) {
        const uint32_t count = level == 0 ? num_blocks : (num_blocks + bpp - 1u) / bpp;
        const uint32_t per = (count + 1023u) / 1024u;
        const uint32_t e0 = min(count, threadIdx.x * per), e1 = min(count, e0 + per);
        auto is_live = [&](uint32_t e) -> bool {
            if (level == 0) return skip[e] == 0u;
            bool any = false;
            for (uint32_t b = e * bpp; b < min(num_blocks, (e + 1u) * bpp); ++b) any = any || skip[b] == 0u;
            return any;
        };
        uint32_t mine = 0;
        for (uint32_t e = e0; e < e1; ++e) mine += is_live(e) ? 1u : 0u;
        uint32_t incl = wave_inclusive_scan(mine, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t t = wave_tot[w];
            base += w < wave ? t : 0u;
            total += t;
        }
        uint32_t pos = base + incl - mine;
        for (uint32_t e = e0; e < e1; ++e) {
            if (is_live(e)) out[pos++] = e;
            else if (level == 0) block_sums[e] = make_uint4(0u, 0u, 0u, 1u);  // .w: skipped (debug tap)
        }
        if (threadIdx.x == 0) live[level] = total;
        out += num_blocks;
        __syncthreads();
    }
}

// batched frames: blockIdx.y = frame; the marks of frame f at block_skip[f * batch.blocks ..]
__global__ __launch_bounds__(256) void block_cull_batch_kernel(FrameBatch batch, const float4 *__restrict__ block_bounds,
                                                               uint32_t num_blocks, uint32_t *__restrict__ block_skip) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x, f = blockIdx.y;
    if (b >= num_blocks) return;
    block_skip[f * batch.blocks + b] =
        block_outside(batch.f[f], block_bounds[3 * b], block_bounds[3 * b + 1], block_bounds[3 * b + 2]) ? 1u : 0u;
}

// ---------------------------------------------------------------------------------------------------
// The compositor's tile schedule, built by ONE EXTRA workgroup of the projection launch (the longest launch before the
// compositor: the ~10 us of ordering 8160 tiles hide behind 12 000 projection workgroups; as an extra workgroup of the
// 17 us scan kernel, round 2, they set that kernel's duration).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t ORDER_CLASSES = 32;
// cost class of a tile for the compositor's schedule: 0 = heaviest (half a staging batch per class, capped)
__device__ __forceinline__ uint32_t order_class(uint32_t staged) {
    const uint32_t c = (staged + 127u) >> 7;
    return ORDER_CLASSES - 1u - (c < ORDER_CLASSES - 1u ? c : ORDER_CLASSES - 1u);
}

// The extra workgroup of scan_blocks_kernel.  It adds up what the compositor staged per tile in the PREVIOUS frame (D_c)
// and posts it to host-mapped memory (the host picks the next frame's colour mode from it and the visible count),
// and it orders the stripe's tiles by those counts, heaviest first: the compositor takes its tiles in that order
// (longest-processing-time-first), so the launch ends on cheap tiles instead of on whichever expensive tile happened to
// come last.  A stable counting sort over 32 cost classes without atomics: the classes go to LDS with independent,
// coalesced loads; wave w then owns a contiguous range of the stripe's tiles, counts its classes with ballots (one per
// class PRESENT in a 64-tile step: staged counts are mostly whole batches, so 2-3) and lane c keeps class c's running
// position.  Changes the schedule only, never the image.
// One 64-slot step of a wave's stable counting sort by cost class (0 .. ORDER_CLASSES; ~0u = no slot in this lane):
// the lanes that share a class are found with six ballots (wave64 match-any, as in the sort's ranking — no loop over the
// classes present in the step, whose trip count made the ordering of a busy frame take ~35 us), `cnt` is the wave's
// own row of running per-class counters in LDS (LDS operations of one wave complete in order).
// COUNT_ONLY: first sweep (class totals of the wave's range); otherwise returns the position of this lane's slot.
template <bool COUNT_ONLY>
__device__ __forceinline__ uint32_t order_step(uint32_t cls, int lane, volatile uint32_t *cnt) {
    const bool valid = cls != ~0u;
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        const bool bit = (cls >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    uint32_t pos = 0;
    if (valid) {
        const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const bool last = (m >> lane) <= 1ull;  // highest lane of the group
        const uint32_t before = cnt[cls];
        pos = before + below;
        if (last) cnt[cls] = before + below + 1u;
    }
    return COUNT_ONLY ? 0u : pos;
}

// what the extra workgroup of the projection launch is handed (all device pointers; order == nullptr: no table)
struct ScheduleArgs {
    const uint32_t *tile_staged;   // per tile: pairs the compositor staged in the PREVIOUS frame of this context
    uint32_t num_tiles;
    uint32_t *dc_parts;            // device, 8 words: the previous frame's D_c, one part per schedule workgroup
                                   // (scan_blocks_kernel adds them up and posts the sum to the host)
    uint32_t *tile_order;
    uint32_t order_mode;
    uint32_t xcd_blocks;           // 1: XCD x projects the contiguous eighth of the 512-slot blocks (project_kernel)
};
constexpr size_t SCHEDULE_LDS_BYTES = ORDER_MAX_SLOTS + 16 * (ORDER_CLASSES + 1) * sizeof(uint32_t) + 16 * sizeof(uint32_t);

// NW wave64 of one workgroup; lds: SCHEDULE_LDS_BYTES of the workgroup's shared memory.  ORDER_XCD: workgroup `part` of
// 8 orders XCD `part`'s list (the lists are independent: eight workgroups, each with a CU to itself, take ~5 us where one
// took ~40 — a lone wave per SIMD runs every instruction at full latency); otherwise one workgroup does everything.
template <int NW>
__device__ __forceinline__ void schedule_tiles(const uint32_t *__restrict__ tile_staged, uint32_t num_tiles,
                                               uint32_t *__restrict__ dc_parts, uint32_t *__restrict__ tile_order,
                                               uint32_t order_mode, uint32_t sx0, uint32_t sx1, uint32_t sy0,
                                               uint32_t sy1, uint32_t gx, uint8_t *lds, uint32_t part) {
    constexpr uint32_t NT = NW * 64u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t dc_prev = 0;
    if (tile_order == nullptr)
        for (uint32_t t = threadIdx.x; t < num_tiles; t += NT) dc_prev += tile_staged[t];
    uint8_t *cls_of = lds;
    uint32_t(*cls_base)[ORDER_CLASSES + 1] = reinterpret_cast<uint32_t(*)[ORDER_CLASSES + 1]>(lds + ORDER_MAX_SLOTS);
    uint32_t *dc_s = reinterpret_cast<uint32_t *>(lds + ORDER_MAX_SLOTS + 16 * (ORDER_CLASSES + 1) * sizeof(uint32_t));
    const uint32_t sw = sx1 - sx0;
    if (tile_order != nullptr && order_mode == ORDER_LPT) {
        const uint32_t stripe_tiles = sw * (sy1 - sy0);
#pragma unroll 8
        for (uint32_t t = threadIdx.x; t < stripe_tiles; t += NT) {
            const uint32_t st = tile_staged[(sy0 + t / sw) * gx + sx0 + t % sw];
            dc_prev += st;  // tiles outside the stripe stage nothing
            cls_of[t] = (uint8_t)order_class(st);
        }
        __syncthreads();
        const uint32_t per_wave = ((stripe_tiles + NT - 1u) / NT) * 64u;
        const uint32_t w_begin = min(stripe_tiles, (uint32_t)wave * per_wave), w_end = min(stripe_tiles, w_begin + per_wave);
        if (lane <= (int)ORDER_CLASSES) cls_base[wave][lane] = 0u;  // this wave's counters: tiles per class
        for (uint32_t t0 = w_begin; t0 < w_end; t0 += 64u) {
            const uint32_t t = t0 + lane;
            (void)order_step<true>(t < w_end ? cls_of[t] : ~0u, lane, cls_base[wave]);
        }
        __syncthreads();
        // position of (wave, class) = tiles of heavier classes + tiles of this class in earlier waves: lane c of the
        // first wave walks class c down the 16 waves, the class totals are scanned across its lanes
        if (wave == 0) {
            uint32_t within[NW], total = 0;
            const int c = lane & (int)(ORDER_CLASSES - 1u);
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                within[w] = total;
                total += cls_base[w][c];
            }
            uint32_t incl_c = lane < (int)ORDER_CLASSES ? total : 0u;
#pragma unroll
            for (int d = 1; d < (int)ORDER_CLASSES; d <<= 1) {
                const uint32_t u = __shfl_up(incl_c, d, 64);
                if (lane >= d) incl_c += u;
            }
            const uint32_t class_first = incl_c - total;
            if (lane < (int)ORDER_CLASSES) {
#pragma unroll
                for (int w = 0; w < NW; ++w) cls_base[w][c] = class_first + within[w];
            }
        }
        __syncthreads();
        for (uint32_t t0 = w_begin; t0 < w_end; t0 += 64u) {
            const uint32_t t = t0 + lane;
            const uint32_t pos = order_step<false>(t < w_end ? cls_of[t] : ~0u, lane, cls_base[wave]);
            if (t < w_end) tile_order[pos] = (sy0 + t / sw) * gx + sx0 + t % sw;
        }
    } else if (tile_order != nullptr) {
        // ORDER_XCD: slot e = x * per_xcd + j enumerates XCD x's tiles block after block (tiles of a block column-major:
        // vertical neighbours first); this workgroup's NW waves order XCD `part`'s list — the same stable counting sort,
        // with class ORDER_CLASSES (= lighter than everything) for the empty slots of partial and virtual blocks
        const OrderLayout lay = order_layout(sw, sy1 - sy0);
        const uint32_t bsz = lay.bw * lay.bh;
        // slot j of XCD x -> tile id (~0u: empty slot).  Divisions by the float reciprocal (operands < 2^24, corrected by
        // at most one): an integer divide is ~40 instructions, and this workgroup has a CU to itself — every instruction
        // at full latency
        const float inv_bsz = 1.0f / (float)bsz, inv_nbx = 1.0f / (float)lay.nbx;
        auto fdiv = [](uint32_t a, uint32_t b, float inv_b) -> uint32_t {
            uint32_t q = (uint32_t)((float)a * inv_b);
            if (q * b > a) --q;
            if ((q + 1u) * b <= a) ++q;
            return q;
        };
        auto tile_of_xj = [&](uint32_t x, uint32_t j) -> uint32_t {
            const uint32_t q = fdiv(j, bsz, inv_bsz), sl = j - q * bsz, B = x + 8u * q;
            const uint32_t by_ = fdiv(B, lay.nbx, inv_nbx), bx_ = B - by_ * lay.nbx;
            const uint32_t sx = lay.bh == 2u ? sl >> 1 : sl, sy = lay.bh == 2u ? sl & 1u : 0u;
            const uint32_t tx = sx0 + bx_ * lay.bw + sx, ty = sy0 + by_ * lay.bh + sy;
            return (B < lay.nblocks && tx < sx1 && ty < sy1) ? ty * gx + tx : ~0u;
        };
        // every wave classifies the slots of the piece of the list it will order (the same slots in all three sweeps)
        const uint32_t xcd = part;
        const uint32_t per_piece = ((lay.per_xcd + 64u * NW - 1u) / (64u * NW)) * 64u;
        const uint32_t j_begin = min(lay.per_xcd, (uint32_t)wave * per_piece), j_end = min(lay.per_xcd, j_begin + per_piece);
#pragma unroll 4
        for (uint32_t j = j_begin + (uint32_t)lane; j < j_end; j += 64u) {
            const uint32_t t = tile_of_xj(xcd, j);
            const uint32_t st = t != ~0u ? tile_staged[t] : 0u;
            dc_prev += st;
            cls_of[j] = (uint8_t)(t != ~0u ? order_class(st) : ORDER_CLASSES);
        }
        if (lane <= (int)ORDER_CLASSES) cls_base[wave][lane] = 0u;  // this wave's counters: slots per class
        __syncthreads();
        for (uint32_t j0 = j_begin; j0 < j_end; j0 += 64u) {
            const uint32_t j = j0 + lane;
            (void)order_step<true>(j < j_end ? cls_of[j] : ~0u, lane, cls_base[wave]);
        }
        __syncthreads();
        {   // every wave: first position of (class, this wave's piece) = slots of heavier classes in all pieces + slots of
            // this class in the earlier pieces
            const bool has = lane <= (int)ORDER_CLASSES;
            uint32_t total = 0, before = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t c = has ? cls_base[w][lane] : 0u;
                before += w < wave ? c : 0u;
                total += c;
            }
            uint32_t incl_c = total;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(incl_c, d, 64);
                if (lane >= d) incl_c += u;
            }
            const uint32_t first = incl_c - total + before;
            __syncthreads();  // (every wave has read every piece's totals)
            if (has) cls_base[wave][lane] = first;  // the counters now run from there
        }
        for (uint32_t j0 = j_begin; j0 < j_end; j0 += 64u) {
            const uint32_t j = j0 + lane;
            const uint32_t pos = order_step<false>(j < j_end ? cls_of[j] : ~0u, lane, cls_base[wave]);
            if (j < j_end) tile_order[pos * 8u + xcd] = tile_of_xj(xcd, j);
        }
    }
    if (dc_parts != nullptr) {
        __syncthreads();
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) dc_prev += __shfl_xor(dc_prev, d, 64);
        if (lane == 0) dc_s[wave] = dc_prev;
        __syncthreads();
        dc_prev = 0;
        for (int w = 0; w < NW; ++w) dc_prev += dc_s[w];
        if (threadIdx.x == 0) dc_parts[part] = dc_prev;
    }
}

// ---------------------------------------------------------------------------------------------------
// project_kernel: one lane per storage slot.
// ---------------------------------------------------------------------------------------------------
// BATCH (batched frames, gsplat_internal.h FrameBatch): `fp` is the VIRTUAL frame (what the schedule workgroups order),
// num_blocks counts VIRTUAL workgroups: workgroup v = frame v / batch.blocks, slots of block v % batch.blocks, projected with
// that frame's camera; every output is indexed by v resp. by the virtual slot v * 512 + lane, and the rectangle's origin
// moves to the frame's rows of the virtual grid.
template <int EAGER, bool BATCH = false>
__global__ __launch_bounds__(PROJ_BLOCK) void project_kernel(SceneSoA scene, uint32_t n, FrameParams fp,
                                                             float4 *__restrict__ culled, SplatKeys keys,
                                                             uint4 *__restrict__ block_sums,
                                                             uint32_t *__restrict__ splat_hist, uint32_t hist_stride,
                                                             const uint32_t *__restrict__ block_skip,
                                                             uint32_t num_blocks, ScheduleArgs sched,
                                                             std::conditional_t<BATCH, FrameBatch, NoBatch> batch,
                                                             const uint32_t *__restrict__ live) {
    __shared__ uint32_t wave_tot[PROJ_BLOCK / 64];
    __shared__ uint32_t wave_vis[PROJ_BLOCK / 64];
    __shared__ uint32_t wave_last[PROJ_BLOCK / 64];
    __shared__ uint32_t hist[256];  // (depth16 & 255) of the workgroup's visible splats: pass 0 of the splat sort
    // a wave's 64 records on their way out (below); the schedule workgroup's scratch
    static_assert(sizeof(float4) * (PROJ_BLOCK / 64) * 64 * 3 >= SCHEDULE_LDS_BYTES, "the schedule borrows the staging array");
    __shared__ float4 stage[PROJ_BLOCK / 64][64 * 3];
    // The extra workgroup (the compositor's tile schedule of this frame) is workgroup 0: the dispatcher starts the
    // workgroups in index order, so its ~20 us run beside the first projection workgroups instead of after the last
    // (as the launch's last workgroup it lengthened the kernel by its whole duration: measured, +21 us).
    // Which 512 slots a projection workgroup takes.  The dispatcher places workgroup b on XCD b % 8 and every XCD has its
    // own L2; each workgroup ends with 256 scattered 4-byte writes, one per row of splat_hist, at column `block`.  With
    // block = b those of NEIGHBOURING columns came from eight different L2s and every one of them left as a partial
    // 32-byte sector: 96 MB of HBM writes per frame for 12 MB of histogram at 6 M splats, in a kernel that runs at the
    // HBM rate.  XCD x takes the contiguous eighth [x per, (x + 1) per) of the blocks instead, in ascending order over
    // its workgroups (the sort's PartitionWalk, same reason), so the entries of a 128-byte line meet in one L2
    // (sched.xcd_blocks; GSPLAT_PROJ_ORDER=linear keeps block = b for A/B).
    const uint32_t per_xcd = (num_blocks + 7u) >> 3;
    const uint32_t extra = gridDim.x - (sched.xcd_blocks ? 8u * per_xcd : num_blocks);  // 0, 1 or 8 (one per XCD list)
    if (blockIdx.x < extra) {
        schedule_tiles<PROJ_BLOCK / 64>(sched.tile_staged, sched.num_tiles, sched.dc_parts, sched.tile_order,
                                        sched.order_mode, fp.sx0, fp.sx1, fp.sy0, fp.sy1, fp.gx,
                                        reinterpret_cast<uint8_t *>(&stage[0][0]), blockIdx.x);
        return;
    }
    const uint32_t b = blockIdx.x - extra;
    uint32_t block;  // the 512 slots this workgroup projects
    if (live != nullptr) {
        // block culling ran: XCD x takes the contiguous eighth x of the LIVE blocks (live_lists_kernel), nobody visits a
        // skipped one
        const uint32_t nlive = live[0];
        const uint32_t per_live = (nlive + 7u) >> 3;
        const uint32_t idx = sched.xcd_blocks ? (b & 7u) * per_live + (b >> 3) : b;
        if ((sched.xcd_blocks && (b >> 3) >= per_live) || idx >= nlive) return;
        block = live[LIVE_HEADER + idx];
    } else {
        block = sched.xcd_blocks ? (b & 7u) * per_xcd + (b >> 3) : b;
        if (block >= num_blocks) return;  // (8 per_xcd >= num_blocks: the last XCD's share may be short)
    }
    // the frame this workgroup projects for, the scene slot of this lane (id) and where its outputs go (vid)
    uint32_t frame = 0, slot_block = block;
    if constexpr (BATCH) {
        frame = block / batch.blocks;
        slot_block = block - frame * batch.blocks;
    }
    const FrameParams &rf = frame_of<BATCH>(fp, batch, frame);
    const uint32_t id = slot_block * PROJ_BLOCK + threadIdx.x;
    const uint32_t vid = block * PROJ_BLOCK + threadIdx.x;
    if (live == nullptr && block_skip != nullptr && block_skip[block]) {  // workgroup-uniform (block_cull_kernel)
        // A skipped workgroup writes 16 bytes and leaves.  (Round 4's wrote its 512 zero rectangle sizes and its 256
        // histogram entries — one scattered 4-byte store per row of splat_hist — so that the splat sort would find no
        // element: 3 KiB per skipped block, two thirds of the blocks on a stripe rank.  The readers look at block_skip
        // themselves now: spine_kernel, downsweep_splats_kernel<true>, the counts tap.)
        if (threadIdx.x == 0) block_sums[block] = make_uint4(0u, 0u, 0u, 1u);  // .w: skipped (debug tap)
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t key = 0, dims = 0, last_plus1 = 0;
    float4 record[3] = {make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f),
                        make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    // EAGER >= 0: RasterizeData with colours; -1: a lazy frame writes no record at all (its compositor works from the
    // scene); -2: a geometry-eager lazy frame — the 32-byte STAGED geometry of every visible splat (project_math.h
    // staged_geometry: centre, pre-multiplied conic, opacity), colours left to the compositor
    constexpr bool GEO = EAGER == -2;
    constexpr bool RECORD = EAGER >= 0 || GEO;
    constexpr int PER = GEO ? 2 : 3;   // float4 per slot of the record buffer
    const uint32_t count = project_splat<(GEO ? -1 : EAGER), RECORD>(scene, n, rf, id, record, key, dims, last_plus1);
    if constexpr (BATCH) {
        // the rectangle's origin tile in the virtual grid: row ty of the real stripe -> frame * rows + (ty - sy0); the host
        // has checked that virtual tile ids fit the key's 16 bits
        key += (((frame * batch.rows - rf.sy0) * rf.gx) & 0xFFFFu) << 16;
    }
    if constexpr (GEO) {
        float4 g0, g1;
        staged_geometry(record[0], record[1], record[2].w, g0, g1);
        record[0] = g0; record[1] = g1;
    }
    // RasterizeData out.  A lane's record is 48 bytes: stored lane by lane, a wave's three stores each touch 64 separate
    // 16-byte pieces at a 48-byte stride.  A wave most of whose splats are visible hands its 64 records through LDS
    // instead and writes 3 x 1 KiB contiguous (the records of its invisible lanes go out as zeros: nobody reads them);
    // sparse waves (a stripe rank) keep the direct stores.  Kernel -10 % at 6 M splats; the HBM write traffic is the same
    // (the L2 merged the pieces before): fewer, whole-line store instructions.
    if constexpr (RECORD) {
        const unsigned long long vis_now = __ballot(count != 0);
        const uint32_t wave_first = block * PROJ_BLOCK + (uint32_t)wave * 64u;   // (virtual slot)
        if (__popcll(vis_now) >= 48 && (BATCH || wave_first + 64u <= n)) {
            float4 *st = stage[wave];
#pragma unroll
            for (int k = 0; k < PER; ++k) st[lane * PER + k] = record[k];
            // (one wave wrote, the same wave reads other lanes' words: its LDS operations complete in order; the
            // barrier below is for the compiler, which sees no dependence between different addresses of one thread)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float4 *dst = culled + (size_t)wave_first * PER;
#pragma unroll
            for (int k = 0; k < PER; ++k) dst[k * 64 + lane] = st[k * 64 + lane];
        } else if (count) {
            float4 *out = culled + (size_t)vid * PER;
#pragma unroll
            for (int k = 0; k < PER; ++k) out[k] = record[k];
        }
    }
    if (BATCH || id < n) {  // (batch: the hand-off arrays are padded to whole workgroups per frame; slots past N hold 0)
        keys.dims[vid] = dims;
        if (count) {
            keys.key[vid] = key;
            atomicAdd(&hist[key & 255u], 1u);
        }
    }

    // (no global atomics here: ~10^5 waves hitting one counter serialise at ~11 ns each — the per-workgroup pair
    // total, visible count and last tile are reduced by scan_blocks_kernel)
    uint32_t pairs = count;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        pairs += __shfl_xor(pairs, d, 64);
        last_plus1 = max(last_plus1, (uint32_t)__shfl_xor((int)last_plus1, d, 64));
    }
    const unsigned long long vis = __ballot(count != 0);
    if (lane == 0) {
        wave_tot[wave] = pairs;
        wave_vis[wave] = (uint32_t)__popcll(vis);
        wave_last[wave] = last_plus1;
    }
    __syncthreads();
    if (threadIdx.x < 256) splat_hist[(size_t)threadIdx.x * hist_stride + block] = hist[threadIdx.x];
    if (threadIdx.x == 0) {
        uint32_t t = 0, v = 0, l = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) {
            t += wave_tot[w];
            v += wave_vis[w];
            l = max(l, wave_last[w]);
        }
        block_sums[block] = make_uint4(t, v, l, 0u);
    }
}

// parity tap: the RasterizeData record of EVERY visible splat, colour included (a lazy frame writes none: its compositor
// recomputes the geometry and evaluates the colour of the splats it stages, from the scene)
template <int DEG>
__global__ __launch_bounds__(256) void fill_records_kernel(SceneSoA scene, uint32_t n, FrameParams fp,
                                                           float4 *__restrict__ culled) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t key, dims, last;
    float4 record[3];
    const uint32_t count = project_splat<DEG, true>(scene, n, fp, id, record, key, dims, last);
    if (count == 0u) return;
    float4 *r = culled + (size_t)id * 3;
    r[0] = record[0]; r[1] = record[1]; r[2] = record[2];
}

// ---------------------------------------------------------------------------------------------------
// Pairs per 512-splat block of the sorted splat list: num_tiles_touched = w * h summed over the block
// (scan_blocks_kernel turns the totals into bases; emit_kernel recomputes the offsets inside a block).
// ---------------------------------------------------------------------------------------------------
// One WAVE per 512-entry block (a lane reads its 8 entries as two uint4), eight blocks per workgroup, no LDS and no
// barrier: 13 -> ~6 us at 6 M splats (one lane per entry made 12 000 workgroups of 512 single-word loads).
__global__ __launch_bounds__(PROJ_BLOCK) void emit_sums_kernel(SplatList list, const uint32_t *__restrict__ v_count,
                                                               uint32_t num_blocks, uint32_t *__restrict__ emit_sums) {
    const uint32_t v = *v_count;
    const uint32_t block = blockIdx.x * (PROJ_BLOCK / 64) + (threadIdx.x >> 6);
    if (block >= num_blocks) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = block * PROJ_BLOCK + lane * 8u;
    uint32_t count = 0;
    if (block * PROJ_BLOCK < v) {  // wave-uniform
        if (first + 8u <= v) {
            const uint4 a = *reinterpret_cast<const uint4 *>(list.dims + first);
            const uint4 b = *reinterpret_cast<const uint4 *>(list.dims + first + 4u);
            const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) count += (d[e] & 0xFFFFu) * (d[e] >> 16);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 8u; ++e)
                if (first + e < v) {
                    const uint32_t d = list.dims[first + e];
                    count += (d & 0xFFFFu) * (d >> 16);
                }
        }
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) count += __shfl_xor(count, k, 64);
    }
    if (lane == 0u) emit_sums[block] = count;
}

// ---------------------------------------------------------------------------------------------------
// Two-round frames (DESIGN.md §4 "occlusion rounds").  The compositor leaves a tile at the first 256-pair batch
// boundary where every pixel is saturated (gsplat_render.glsl:66,97); the pairs behind that point are sorted by the
// reference and never read.  A two-round frame composites the front part of the depth-sorted splat list first (round
// A: the first plan.v_a list entries), then emits, for the rest of the list (round B), only the splats whose
// rectangle still holds an unfinished tile.  Same pixels, bit for bit: the per-pixel state, the number of pairs a tile
// has consumed and the batch boundaries carry over from A to B (raster.hip).
//
// frame_plan_kernel (one workgroup, after the projection pass): D and V of the whole frame from the projection
// workgroups' records, and the size of round A.  A frame whose D exceeds the key budget is composited in one round
// (plan.single): the reference drops the pairs past the budget in emission order, which only a full emission knows.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void frame_plan_kernel(const uint4 *__restrict__ proj_sums, uint32_t num_blocks,
                                                          uint64_t capacity, uint32_t frac16,
                                                          uint64_t *__restrict__ total_out,
                                                          FramePlan *__restrict__ plan,
                                                          uint32_t *__restrict__ d_hint) {
    __shared__ uint64_t pairs_s[16];
    __shared__ uint32_t vis_s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t pairs = 0;
    uint32_t vis = 0;
#pragma unroll 8
    for (uint32_t i = threadIdx.x; i < num_blocks; i += 1024u) {  // (independent loads: in flight together)
        const uint4 bs = proj_sums[i];
        pairs += bs.x;
        vis += bs.y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        pairs += __shfl_xor(pairs, d, 64);
        vis += __shfl_xor(vis, d, 64);
    }
    if (lane == 0) { pairs_s[wave] = pairs; vis_s[wave] = vis; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t d_full = 0;
        uint32_t v = 0;
        for (int w = 0; w < 16; ++w) { d_full += pairs_s[w]; v += vis_s[w]; }
        const bool single = d_full > capacity || frac16 >= 65536u;
        *total_out = d_full;
        if (d_hint != nullptr) *d_hint = d_full < 0xFFFFFFFFull ? (uint32_t)d_full : 0xFFFFFFFFu;  // host-mapped
        plan->single = single ? 1u : 0u;
        plan->v_a = single ? v : (uint32_t)(((uint64_t)v * frac16) >> 16);
        plan->unfinished = 0u;  // (round A's compositor counts the tiles it leaves unfinished)
    }
}

// Re-laid-out scenes: the sorted list is in (depth16, slot) order and the boundaries pass repairs runs of equal (tile,
// depth16) keys into ascending splat id — a run must not be cut by the end of round A.  One workgroup moves plan.v_a
// forward to the next entry whose depth code differs from its predecessor's (or to the end of the list).
__global__ __launch_bounds__(1024) void plan_align_kernel(const uint32_t *__restrict__ list_key,
                                                          const uint32_t *__restrict__ v_count,
                                                          FramePlan *__restrict__ plan) {
    __shared__ uint32_t found;
    const uint32_t v = *v_count;
    uint32_t start = plan->v_a;
    if (plan->single || start == 0u || start >= v) {
        if (threadIdx.x == 0 && start > v) plan->v_a = v;
        return;
    }
    if (threadIdx.x == 0) found = v;
    __syncthreads();
    for (uint32_t base = start; base < v; base += 1024u) {
        const uint32_t j = base + threadIdx.x;
        if (j < v && (list_key[j] & 0xFFFFu) != (list_key[j - 1u] & 0xFFFFu)) atomicMin(&found, j);
        __syncthreads();
        if (found < v) break;  // (uniform: read after the barrier)
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) plan->v_a = found;
}

// Summed-area table of the tiles round A left unfinished (u16, (gy + 1) x (gx + 1), row and column 0 zero): a
// rectangle of tiles holds an unfinished one iff its four-corner sum is non-zero.  One workgroup, the table lives
// in LDS while it is built (at most 32 768 tiles: api.hip).  Tiles outside the stripe count as finished (nothing is
// emitted for them).
__global__ __launch_bounds__(1024) void tile_sat_kernel(const uint32_t *__restrict__ tile_done,
                                                        const FramePlan *__restrict__ plan, uint32_t gx, uint32_t gy,
                                                        uint32_t sx0, uint32_t sx1, uint32_t sy0, uint32_t sy1,
                                                        uint16_t *__restrict__ sat) {
    extern __shared__ uint16_t sat_s[];
    const uint32_t pitch = gx + 1u, entries = pitch * (gy + 1u);
    // round A finished every tile (a dense scene) — or was the whole frame: the table's last entry, its total, is all
    // anybody reads (round_filter_kernel leaves on it), and it is zero
    if (plan->unfinished == 0u) {
        if (threadIdx.x == 0) sat[entries - 1u] = 0u;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t e = threadIdx.x; e < entries; e += 1024u) sat_s[e] = 0;
    __syncthreads();
    {   // the stripe's tiles, row-major over the stripe: independent loads, one division per element by a constant width
        const uint32_t sw = sx1 - sx0, stripe_tiles = sw * (sy1 - sy0);
#pragma unroll 8
        for (uint32_t t = threadIdx.x; t < stripe_tiles; t += 1024u) {
            const uint32_t ty = sy0 + t / sw, tx = sx0 + t % sw;
            sat_s[(ty + 1u) * pitch + tx + 1u] = tile_done[ty * gx + tx] ? 0u : 1u;
        }
    }
    __syncthreads();
    for (uint32_t r = 1u + (uint32_t)wave; r <= gy; r += 16u) {  // along the rows: one wave per row, 64 columns a step
        uint32_t carry = 0;
        for (uint32_t c0 = 1u; c0 <= gx; c0 += 64u) {
            const uint32_t c = c0 + (uint32_t)lane;
            uint32_t v = c <= gx ? sat_s[r * pitch + c] : 0u;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t u = __shfl_up(v, d, 64);
                if (lane >= d) v += u;
            }
            v += carry;
            if (c <= gx) sat_s[r * pitch + c] = (uint16_t)v;
            carry = __shfl(v, 63, 64);
        }
    }
    __syncthreads();
    for (uint32_t c = 1u + threadIdx.x; c <= gx; c += 1024u) {  // down the columns: one lane per column
        uint32_t run = 0;
        for (uint32_t r = 1u; r <= gy; ++r) {
            run += sat_s[r * pitch + c];
            sat_s[r * pitch + c] = (uint16_t)run;
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < entries; e += 1024u) sat[e] = sat_s[e];
}

// Round B's counterpart of emit_sums_kernel, over the WHOLE sorted list.  Entry i < plan.v_a (composited by round
// A): if round A could not finish the frame's last tile T - 1 (the reference drops the LAST pair of that tile's
// complete list, quirk Q6: raster.hip), round B composites T - 1 from scratch, so a splat that covers it emits that one
// pair again.  Entry i >= v_a: all its pairs if its rectangle holds an unfinished tile, none otherwise.  Nothing
// unfinished at all (a dense scene): every workgroup leaves after one read.  The effective rectangle goes to out
// (key = depth16 | origin tile << 16, dims), which round B's emit_kernel reads in place of the list's.
// (One workgroup per block of the list, or — grid smaller than the number of blocks — a walk over the blocks: round B of a
// dense scene has nothing to do, and N/512 workgroups that leave at once still cost their dispatch.)
__global__ __launch_bounds__(PROJ_BLOCK) void round_filter_kernel(SplatList list, const uint32_t *__restrict__ v_count,
                                                                  const FramePlan *__restrict__ plan,
                                                                  const uint16_t *__restrict__ sat,
                                                                  const uint32_t *__restrict__ tile_done, uint32_t gx,
                                                                  uint32_t gy, uint32_t *__restrict__ key_out,
                                                                  uint32_t *__restrict__ dims_out,
                                                                  uint32_t *__restrict__ emit_sums, uint32_t num_blocks) {
    __shared__ uint32_t wave_tot[2][PROJ_BLOCK / 64];  // (two rows: a fast wave may be one block ahead of a slow one)
    const uint32_t v = *v_count;
    const bool nothing = sat[(gy + 1u) * (gx + 1u) - 1u] == 0u;
    uint32_t it = 0;  // (counts the blocks that reach the barrier below: the LDS row alternates with THOSE)
    for (uint32_t blk = blockIdx.x; blk < num_blocks; blk += gridDim.x) {
        const uint32_t i = blk * PROJ_BLOCK + threadIdx.x;
        if (blk * PROJ_BLOCK >= v || nothing) {  // workgroup-uniform
            if (threadIdx.x == 0) emit_sums[blk] = 0u;
            continue;
        }
        const bool redo_last_tile = tile_done[gx * gy - 1u] == 0u;
        uint32_t count = 0, eff_key = 0, eff_dims = 0;
        if (i < v) {
            uint32_t key = list.key[i], d = list.dims[i];
            const uint32_t w = d & 0xFFFFu, h = d >> 16, t0 = key >> 16;
            const uint32_t y0 = t0 / gx, x0 = t0 - y0 * gx;
            if (plan->single) {
                d = 0u;
            } else if (i < plan->v_a) {
                if (redo_last_tile && w != 0u && x0 + w == gx && y0 + h == gy) {
                    key = (key & 0xFFFFu) | ((gx * gy - 1u) << 16);
                    d = 1u | (1u << 16);
                } else {
                    d = 0u;
                }
            } else {
                const uint32_t pitch = gx + 1u;
                const uint32_t s = (uint32_t)sat[(y0 + h) * pitch + x0 + w] - (uint32_t)sat[y0 * pitch + x0 + w] -
                                   (uint32_t)sat[(y0 + h) * pitch + x0] + (uint32_t)sat[y0 * pitch + x0];
                if ((s & 0xFFFFu) == 0u) d = 0u;
            }
            eff_key = key;
            eff_dims = d;
            count = (d & 0xFFFFu) * (d >> 16);
        }
#pragma unroll
        for (int k = 32; k >= 1; k >>= 1) count += __shfl_xor(count, k, 64);
        uint32_t *wt = wave_tot[it++ & 1u];
        if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = count;
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) total += wt[w];
        if (threadIdx.x == 0) emit_sums[blk] = total;
        // emit_kernel leaves a block whose total is zero without reading its entries: nothing to write for those
        if (total != 0u && i < v) {
            key_out[i] = eff_key;
            dims_out[i] = eff_dims;
        }
    }
}

// Exclusive scan of the workgroup totals of emit_sums_kernel (N/512 entries); 64-bit bases so a pathological D
// cannot wrap.  Also reduces the visible count and the frame's last tile from the projection workgroups' records,
// finalises D / min(D, capacity) / overflow and clears tile_bounds.
// One workgroup per 1024 totals, no inter-workgroup dependency: workgroup k first reduces ALL totals before its
// slice (k x 4 KiB of reads), then scans its own 1024.  The last workgroup sees every total and writes the counters.
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const uint32_t *__restrict__ emit_sums,
                                                           const uint4 *__restrict__ proj_sums, uint32_t num_blocks,
                                                           uint64_t *__restrict__ block_base, uint64_t capacity,
                                                           uint64_t *__restrict__ total_out,
                                                           uint32_t *__restrict__ d_sorted,
                                                           uint32_t *__restrict__ overflow,
                                                           uint32_t *__restrict__ visible_out,
                                                           uint32_t *__restrict__ last_tile_out,
                                                           uint4 *__restrict__ bounds_as_uint4, uint32_t bounds_uint4s,
                                                           uint32_t *__restrict__ big_count,
                                                           uint32_t *__restrict__ host_hint,
                                                           const uint32_t *__restrict__ dc_parts,
                                                           uint32_t *__restrict__ pairs_hint,
                                                           uint32_t *__restrict__ last_tile_copy,
                                                           uint32_t *__restrict__ long_count,
                                                           uint32_t *__restrict__ big_seen, uint32_t frame_blocks) {
    __shared__ uint64_t wave_pre[16], wave_own[16];
    __shared__ uint32_t vis_s[16], last_s[16];
    __shared__ uint32_t last_f[MAX_BATCH];  // batched frames: the last tile of every frame (LDS atomics: one workgroup, once)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // gaussian_splatting_rasterizer.gd:128 buffer_clear(tile_bounds): done here (a few KiB..260 KiB) instead of a
    // separate fill launch; boundaries_kernel runs after the whole sort, long after this
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < bounds_uint4s; i += gridDim.x * 1024u)
        bounds_as_uint4[i] = make_uint4(0u, 0u, 0u, 0u);

    const bool last_wg = blockIdx.x == gridDim.x - 1;
    const uint32_t first = blockIdx.x * 1024u;
    uint64_t pre = 0;  // pairs of the workgroups before this slice
#pragma unroll 8  // (independent loads: in flight together — the kernel is a handful of dependent round trips long)
    for (uint32_t i = threadIdx.x; i < first; i += 1024u) pre += emit_sums[i];
    uint32_t vis = 0, last = 0;
    if (last_wg) {
        if (frame_blocks != 0u && threadIdx.x < MAX_BATCH) last_f[threadIdx.x] = 0u;
        if (frame_blocks != 0u) __syncthreads();  // (uniform: a kernel argument)
#pragma unroll 8
        for (uint32_t i = threadIdx.x; i < num_blocks; i += 1024u) {
            const uint4 bs = proj_sums[i];
            vis += bs.y;
            last = max(last, bs.z);
            if (frame_blocks != 0u && bs.z != 0u) atomicMax(&last_f[i / frame_blocks], bs.z);
        }
    }
    const uint32_t i = first + threadIdx.x;
    const uint32_t own = i < num_blocks ? emit_sums[i] : 0u;
    uint64_t incl = own;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        pre += __shfl_xor(pre, d, 64);
        vis += __shfl_xor(vis, d, 64);
        last = max(last, (uint32_t)__shfl_xor((int)last, d, 64));
    }
    if (lane == 63) wave_own[wave] = incl;
    if (lane == 0) { wave_pre[wave] = pre; vis_s[wave] = vis; last_s[wave] = last; }
    __syncthreads();
    uint64_t base = 0, own_total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        base += wave_pre[w];
        const uint64_t t = wave_own[w];
        if (w < wave) base += t;
        own_total += t;
    }
    if (i < num_blocks) block_base[i] = base + incl - own;
    if (last_wg && threadIdx.x == 0) {
        uint64_t total = own_total;
        uint32_t vv = 0, l = 0;
        for (int w = 0; w < 16; ++w) { total += wave_pre[w]; vv += vis_s[w]; l = max(l, last_s[w]); }
        *total_out = total;
        *d_sorted = (uint32_t)(total < capacity ? total : capacity);
        if (pairs_hint != nullptr) *pairs_hint = (uint32_t)(total < capacity ? total : capacity);  // host-mapped
        *overflow = total > capacity ? 1u : 0u;
        *visible_out = vv;
        if (frame_blocks == 0u) {
            *last_tile_out = l;
            // (gsplat_render_begin's caller gets the word here: round 3 issued a 4-byte device-to-device copy for it, a blit
            // kernel launch per frame)
            if (last_tile_copy != nullptr) *last_tile_copy = l;
        } else {  // one word per frame of the batch (the barrier after the reduction above ordered the LDS atomics)
            for (int f = 0; f < MAX_BATCH; ++f) {
                last_tile_out[f] = last_f[f];
                if (last_tile_copy != nullptr) last_tile_copy[f] = last_f[f];
            }
        }
        // big rectangles the emissions since the last posting met (a frame's round B and replays scan without posting)
        const uint32_t big_prev = max(*big_seen, *big_count);
        *big_seen = host_hint != nullptr ? 0u : big_prev;
        *big_count = 0u;  // emit_kernel's list of big rectangles starts empty
        *long_count = 0u; // ... and so does the boundaries pass's list of long runs of equal keys (finalized scenes)
        if (host_hint != nullptr) {
            host_hint[3] = big_prev;
            host_hint[0] = vv;
            uint32_t dc_prev = 0;  // D_c of the previous frame: the schedule workgroups of this frame's projection launch
            for (int k = 0; k < 8; ++k) dc_prev += dc_parts[k];
            host_hint[1] = dc_prev;
            host_hint[2] = ++big_count[3];  // frames posted so far, counted in device memory (Counters::hint_frames)
        }
    }
}

// gsplat_projection.glsl:218-226: duplicate (key, id) over the tile rectangle, y outer / x inner.
// The slot of every pair is fixed by the scans (block_base + offset within the block (a workgroup scan of
// num_tiles_touched = the deterministic stand-in for the atomicAdd of :196) + y-outer/x-inner index), so any distribution of
// the writes gives the same buffers; two levels keep the stores coalesced and the load balanced whatever the splat sizes:
//  * emit_kernel — wave-cooperative: the pairs of a wave's 64 splats are numbered 0..T-1 (wave scan), lane l writes
//    pairs l, l+64, ... and finds the owning splat by a 6-step binary search over the lanes' inclusive ends (shuffles).
//  * splats covering more than EMIT_BIG tiles are only *listed* there and written by emit_big_kernel, where the whole
//    grid shares each rectangle.  (A per-lane loop over its own rectangle made 2000 screen-filling splats cost 1.2 ms,
//    and any per-wave scheme still serialises when such splats sit next to each other in the list.
//    tools/big_splats.py is the stress case.)
// (A no-wait look-back over the workgroup totals inside this kernel was tried instead of scan_blocks_kernel: with
// ~2000 workgroups in flight nobody has published a prefix nearby, every workgroup walks ~2000 entries, 2.5x slower.)
constexpr uint32_t EMIT_BIG = 512;

// KeyT = uint16_t: the key is the tile id alone (narrow-key frames, sort.hip)
template <typename KeyT>
__device__ __forceinline__ void write_pair(uint32_t j, uint32_t x0, uint32_t y0, uint32_t wx, uint32_t depth, uint32_t id,
                                           const TileMap &map, uint64_t off, uint64_t capacity, KeyT *__restrict__ keys,
                                           uint32_t *__restrict__ values) {
    // j / wx without an integer divide: float estimate (j < 2^24), corrected by at most one
    uint32_t q = (uint32_t)((float)j * (1.0f / (float)wx));
    int32_t rem = (int32_t)(j - q * wx);
    if (rem < 0) { --q; rem += (int32_t)wx; }
    if (rem >= (int32_t)wx) { ++q; rem -= (int32_t)wx; }
    if (off < capacity) {  // never write past the key budget
        // 16-bit keys: the tile id inside the context's stripe (TileMap, gsplat_internal.h); 32-bit: the reference's key
        if constexpr (sizeof(KeyT) == 2) keys[off] = (KeyT)map.local_of(x0 + (uint32_t)rem, y0 + q);
        else keys[off] = (KeyT)((((y0 + q) * map.gx + (x0 + (uint32_t)rem)) << 16) | depth);
        values[off] = id;
    }
}

template <typename KeyT>
__global__ __launch_bounds__(PROJ_BLOCK) void emit_kernel(SplatList list, const uint32_t *__restrict__ v_count,
                                                          TileMap map, const uint32_t *__restrict__ emit_sums,
                                                          const uint64_t *__restrict__ block_base, uint64_t capacity,
                                                          KeyT *__restrict__ keys, uint32_t *__restrict__ values,
                                                          uint32_t *__restrict__ big_count,
                                                          uint32_t *__restrict__ big_list, uint32_t list_bigs) {
    __shared__ uint32_t wave_tot[PROJ_BLOCK / 64];
    if (emit_sums[blockIdx.x] == 0u) return;  // past the end of the list
    const uint32_t i = blockIdx.x * PROJ_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valid = i < *v_count;
    uint32_t depth = 0, id = 0, x0 = 0, y0 = 0, wx = 0, count = 0;
    if (valid) {
        const uint32_t key = list.key[i], d = list.dims[i];
        id = list.id[i];
        depth = key & 0xFFFFu;
        const uint32_t t0 = key >> 16;
        y0 = t0 / map.gx;
        x0 = t0 - y0 * map.gx;
        wx = d & 0xFFFFu;
        count = wx * (d >> 16);
    }
    const uint64_t base = block_base[blockIdx.x];
    // offset within the workgroup's range (ascending with the list position)
    const uint32_t incl_all = wave_inclusive_scan(count, lane);
    if (lane == 63) wave_tot[wave] = incl_all;
    __syncthreads();
    uint32_t excl = incl_all - count;
#pragma unroll
    for (int w = 0; w < PROJ_BLOCK / 64; ++w)
        if (w < wave) excl += wave_tot[w];

    // big rectangles: wave-aggregated append to the list (order in the list is irrelevant, slots are fixed).  Without
    // the second launch (list_bigs == 0: no recent frame of this context met one) they are only counted — the host reads
    // the count back and asks for the list in the frames that follow — and written below like any other rectangle.
    const bool big = count > EMIT_BIG && base + excl < capacity && blockIdx.y == 0u;
    const unsigned long long big_mask = __ballot(big);
    if (big_mask) {
        uint32_t first_slot = 0;
        if (lane == (int)__builtin_ctzll(big_mask)) first_slot = atomicAdd(big_count, (uint32_t)__popcll(big_mask));
        first_slot = __shfl(first_slot, (int)__builtin_ctzll(big_mask), 64);
        if (big && list_bigs) {
            const uint32_t e = first_slot + (uint32_t)__popcll(big_mask & ((1ull << lane) - 1ull));
            big_list[2 * e] = i;
            big_list[2 * e + 1] = excl;
        }
    }

    // Most splats of a frame cover one to four tiles (c3: 1.7 on average, c2: 2.3): a lane with that few writes its own
    // pairs — the wave's stores still fall into one contiguous run of slots, a few cache lines merged in the L2 — and only
    // the larger rectangles go through the cooperative walk below (13 cross-lane reads per 64 pairs), which in most
    // waves then has nothing or little left to do.
    const bool own = count <= 4u;
    if (blockIdx.y == 0u) {
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            if (own && j < count) write_pair(j, x0, y0, wx, depth, id, map, base + excl + j, capacity, keys, values);
    }
    const uint32_t small = (own || (list_bigs && count > EMIT_BIG)) ? 0u : count;
    const uint32_t incl = wave_inclusive_scan(small, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0u) return;  // wave-uniform
    const uint32_t pair0 = incl - small;  // number of this lane's first pair within the wave
    // gridDim.y workgroups share a block of the list (round A of a two-round frame is a short list of large splats:
    // one wave per 64 of them would leave most of the chip idle): workgroup y takes every gridDim.y-th 64-pair step
    for (uint32_t p = (uint32_t)lane + 64u * blockIdx.y; p < ((total + 63u) & ~63u); p += 64u * gridDim.y) {
        int lo = 0, hi = 63;  // smallest lane whose inclusive end is > p
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            const uint32_t e = __shfl(incl, mid, 64);
            if (e > p) hi = mid; else lo = mid + 1;
        }
        const int src = lo & 63;
        const uint32_t s_pair0 = __shfl(pair0, src, 64), s_excl = __shfl(excl, src, 64);
        const uint32_t s_x0 = __shfl(x0, src, 64), s_y0 = __shfl(y0, src, 64);
        const uint32_t s_wx = __shfl(wx, src, 64), s_depth = __shfl(depth, src, 64), s_id = __shfl(id, src, 64);
        if (p < total) {
            const uint32_t j = p - s_pair0;  // index inside the splat's rectangle, y outer / x inner
            write_pair(j, s_x0, s_y0, s_wx, s_depth, s_id, map, base + s_excl + j, capacity, keys, values);
        }
    }
}

// grid (EMIT_BIG_X, y): blockIdx.y strides over the listed splats, blockIdx.x over 256-pair pieces of one.  y = 32 (256
// workgroups: the launch is empty or nearly so in most frames of most scenes) up to 1024, following the number of big
// rectangles the context's recent emissions met (the host knows it from the scan's posting): a capture-shaped scene (c3r)
// lists ~10^4 of them per frame and most of its pairs are written HERE — on 256 workgroups that was the frame's longest
// kernel class (2 x 51 us, 0.3 TB/s).
constexpr uint32_t EMIT_BIG_X = 8, EMIT_BIG_Y = 32, EMIT_BIG_Y_MAX = 1024;
static uint32_t emit_big_rows(uint32_t big_hint) {
    uint32_t y = EMIT_BIG_Y;
    while (y < EMIT_BIG_Y_MAX && y < big_hint) y <<= 1;
    return y;
}
template <typename KeyT>
__global__ __launch_bounds__(256) void emit_big_kernel(SplatList list, TileMap map,
                                                       const uint64_t *__restrict__ block_base, uint64_t capacity,
                                                       KeyT *__restrict__ keys, uint32_t *__restrict__ values,
                                                       const uint32_t *__restrict__ big_count,
                                                       const uint32_t *__restrict__ big_list) {
    const uint32_t nb = *big_count;
    for (uint32_t e = blockIdx.y; e < nb; e += gridDim.y) {
        const uint32_t i = big_list[2 * e];
        const uint32_t key = list.key[i], d = list.dims[i], id = list.id[i];
        const uint32_t wx = d & 0xFFFFu, count = wx * (d >> 16), depth = key & 0xFFFFu;
        const uint32_t t0 = key >> 16, y0 = t0 / map.gx, x0 = t0 - y0 * map.gx;
        const uint64_t off0 = block_base[i / PROJ_BLOCK] + big_list[2 * e + 1];
        for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < count; j += gridDim.x * 256u)
            write_pair(j, x0, y0, wx, depth, id, map, off0 + j, capacity, keys, values);
    }
}

// taps of a narrow-key frame: the reference's 32-bit key of every pair, rebuilt from the tile id and the splat's depth16
__global__ __launch_bounds__(256) void widen_keys_kernel(const uint16_t *__restrict__ keys16,
                                                         const uint32_t *__restrict__ values,
                                                         const uint32_t *__restrict__ splat_keys,
                                                         const uint32_t *__restrict__ d_count,
                                                         uint32_t *__restrict__ keys_out, TileMap map) {
    const uint32_t count = *d_count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u)
        keys_out[i] = (map.global_of(keys16[i]) << 16) | (splat_keys[values[i]] & 0xFFFFu);
}

// parity tap: num_tiles_touched per slot
__global__ __launch_bounds__(256) void tile_counts_kernel(const uint32_t *__restrict__ dims,
                                                          uint32_t *__restrict__ counts, uint32_t n,
                                                          const uint32_t *__restrict__ block_skip) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const bool skipped = block_skip != nullptr && block_skip[i / PROJ_BLOCK] != 0u;  // (the frame wrote nothing for this block)
    counts[i] = skipped ? 0u : (dims[i] & 0xFFFFu) * (dims[i] >> 16);
}

}  // namespace

static bool proj_xcd_blocks() {  // GSPLAT_PROJ_ORDER=linear|xcd (A/B; same outputs), read once per process
    static const bool v = [] {
        const char *e = getenv("GSPLAT_PROJ_ORDER");
        return !(e && !strcmp(e, "linear"));
    }();
    return v;
}

void launch_project(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                    const SplatKeys &keys, uint4 *block_sums, uint32_t *splat_hist, const float4 *block_bounds,
                    uint32_t *block_skip, const uint32_t *tile_staged, uint32_t num_tiles, uint32_t *dc_parts,
                    const TileSchedule &sched, hipStream_t s, uint32_t *live, uint32_t blocks_per_part) {
    if (n == 0) return;
    // + the workgroups that build the compositor's tile schedule and add up the previous frame's D_c (schedule_tiles):
    // one per XCD list, or one for the single list / for the sum alone
    const uint32_t extra = tile_staged == nullptr ? 0u : (sched.order != nullptr && sched.mode == ORDER_XCD ? 8u : 1u);
    if (extra == 1u) (void)hipMemsetAsync(dc_parts + 1, 0, 7 * sizeof(uint32_t), s);
    const dim3 grid((n + PROJ_BLOCK - 1) / PROJ_BLOCK), block(PROJ_BLOCK);
    const bool xcd_blocks = proj_xcd_blocks();
    const dim3 launch_grid((xcd_blocks ? 8u * ((grid.x + 7u) / 8u) : grid.x) + extra);
    const ScheduleArgs sa{tile_staged, num_tiles, dc_parts, sched.order, sched.mode, xcd_blocks ? 1u : 0u};
    const bool cull = fp.cull_mode != 0u && block_bounds != nullptr && block_skip != nullptr;
    if (cull)
        hipLaunchKernelGGL(block_cull_kernel, dim3((grid.x + 255u) / 256u), dim3(256), 0, s, fp, block_bounds, grid.x,
                           block_skip);
    const uint32_t *skip = cull ? block_skip : nullptr;
    const uint32_t *live_list = cull ? live : nullptr;
    if (live_list != nullptr)
        hipLaunchKernelGGL(live_lists_kernel, dim3(1), dim3(1024), 0, s, block_skip, grid.x, blocks_per_part, live, block_sums);
#define GSPLAT_LAUNCH_P(E)                                                                                       \
    hipLaunchKernelGGL((project_kernel<E, false>), launch_grid, block, 0, s, scene, n, fp, culled, keys, block_sums,    \
                       splat_hist, grid.x, skip, grid.x, sa, NoBatch{}, live_list)
    switch (sh_degree) {  // -1: colours left to the compositor
        case 0: GSPLAT_LAUNCH_P(0); break;
        case 1: GSPLAT_LAUNCH_P(1); break;
        case 2: GSPLAT_LAUNCH_P(2); break;
        case 3: GSPLAT_LAUNCH_P(3); break;
        case -2: GSPLAT_LAUNCH_P(-2); break;
        default: GSPLAT_LAUNCH_P(-1); break;
    }
#undef GSPLAT_LAUNCH_P
}

void launch_project_batch(const SceneSoA &scene, uint32_t n, const FrameBatch &batch, const FrameParams &fpv, int sh_degree,
                          float4 *records, const SplatKeys &keys, uint4 *block_sums, uint32_t *splat_hist,
                          const float4 *block_bounds, uint32_t *block_skip, const uint32_t *tile_staged, uint32_t num_tiles,
                          uint32_t *dc_parts, const TileSchedule &sched, hipStream_t s, uint32_t *live,
                          uint32_t blocks_per_part) {
    if (n == 0) return;
    const uint32_t extra = tile_staged == nullptr ? 0u : (sched.order != nullptr && sched.mode == ORDER_XCD ? 8u : 1u);
    if (extra == 1u) (void)hipMemsetAsync(dc_parts + 1, 0, 7 * sizeof(uint32_t), s);
    const uint32_t vblocks = batch.count * batch.blocks;
    const dim3 block(PROJ_BLOCK);
    const bool xcd_blocks = proj_xcd_blocks();
    const dim3 launch_grid((xcd_blocks ? 8u * ((vblocks + 7u) / 8u) : vblocks) + extra);
    const ScheduleArgs sa{tile_staged, num_tiles, dc_parts, sched.order, sched.mode, xcd_blocks ? 1u : 0u};
    const bool cull = batch.f[0].cull_mode != 0u && block_bounds != nullptr && block_skip != nullptr;
    if (cull)
        hipLaunchKernelGGL(block_cull_batch_kernel, dim3((batch.blocks + 255u) / 256u, batch.count), dim3(256), 0, s, batch,
                           block_bounds, batch.blocks, block_skip);
    const uint32_t *skip = cull ? block_skip : nullptr;
    const uint32_t *live_list = cull ? live : nullptr;
    if (live_list != nullptr)
        hipLaunchKernelGGL(live_lists_kernel, dim3(1), dim3(1024), 0, s, block_skip, vblocks, blocks_per_part, live, block_sums);
#define GSPLAT_LAUNCH_PB(E)                                                                                            \
    hipLaunchKernelGGL((project_kernel<E, true>), launch_grid, block, 0, s, scene, n, fpv, records, keys, block_sums,     \
                       splat_hist, vblocks, skip, vblocks, sa, batch, live_list)
    switch (sh_degree) {
        case 0: GSPLAT_LAUNCH_PB(0); break;
        case 1: GSPLAT_LAUNCH_PB(1); break;
        case 2: GSPLAT_LAUNCH_PB(2); break;
        case 3: GSPLAT_LAUNCH_PB(3); break;
        case -2: GSPLAT_LAUNCH_PB(-2); break;
        default: GSPLAT_LAUNCH_PB(-1); break;
    }
#undef GSPLAT_LAUNCH_PB
}

void launch_pow02_bits(uint32_t first_bits, uint64_t count, float *out, hipStream_t s) {
    if (!count) return;
    hipLaunchKernelGGL(pow02_bits_kernel, dim3(8192), dim3(256), 0, s, first_bits, count, out);
}

void launch_block_bounds(const SceneSoA &scene, uint32_t n, float4 *block_bounds, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(block_bounds_kernel, dim3((n + PROJ_BLOCK - 1) / PROJ_BLOCK), dim3(PROJ_BLOCK), 0, s, scene, n,
                       block_bounds);
}

void launch_fill_records(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                         hipStream_t s) {
    if (n == 0) return;
    const dim3 grid((n + 255u) / 256u), block(256);
    switch (sh_degree <= 0 ? 0 : (sh_degree > 3 ? 3 : sh_degree)) {
        case 0: hipLaunchKernelGGL(fill_records_kernel<0>, grid, block, 0, s, scene, n, fp, culled); break;
        case 1: hipLaunchKernelGGL(fill_records_kernel<1>, grid, block, 0, s, scene, n, fp, culled); break;
        case 2: hipLaunchKernelGGL(fill_records_kernel<2>, grid, block, 0, s, scene, n, fp, culled); break;
        default: hipLaunchKernelGGL(fill_records_kernel<3>, grid, block, 0, s, scene, n, fp, culled); break;
    }
}

void launch_emit_sums(const SplatList &list, const uint32_t *v_count, uint32_t n, uint32_t *emit_sums, hipStream_t s) {
    if (n == 0) return;
    const uint32_t num_blocks = (n + PROJ_BLOCK - 1) / PROJ_BLOCK;
    hipLaunchKernelGGL(emit_sums_kernel, dim3((num_blocks + PROJ_BLOCK / 64 - 1) / (PROJ_BLOCK / 64)), dim3(PROJ_BLOCK), 0, s,
                       list, v_count, num_blocks, emit_sums);
}

void launch_frame_plan(const uint4 *proj_sums, uint32_t num_blocks, uint64_t capacity, uint32_t frac16,
                       uint64_t *total_out, FramePlan *plan, uint32_t *d_hint, hipStream_t s) {
    hipLaunchKernelGGL(frame_plan_kernel, dim3(1), dim3(1024), 0, s, proj_sums, num_blocks, capacity, frac16, total_out, plan,
                       d_hint);
}

void launch_plan_align(const uint32_t *list_key, const uint32_t *v_count, FramePlan *plan, hipStream_t s) {
    hipLaunchKernelGGL(plan_align_kernel, dim3(1), dim3(1024), 0, s, list_key, v_count, plan);
}

size_t tile_sat_entries(uint32_t gx, uint32_t gy) { return (size_t)(gx + 1u) * (gy + 1u); }

int launch_tile_sat(const uint32_t *tile_done, const FramePlan *plan, const FrameParams &fp, uint16_t *sat, hipStream_t s) {
    const size_t bytes = tile_sat_entries(fp.gx, fp.gy) * sizeof(uint16_t);
    // more than the default dynamic-LDS limit has to be asked for once per kernel (the attribute is per kernel, not per
    // stream); contexts on several threads may get here together: setting it twice is harmless, the maximum only grows
    static std::atomic<size_t> allowed{0};
    if (bytes > allowed.load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(tile_sat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ROUNDS_MAX_SAT_BYTES) != hipSuccess)
            return -1;
        allowed.store(ROUNDS_MAX_SAT_BYTES, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(tile_sat_kernel, dim3(1), dim3(1024), bytes, s, tile_done, plan, fp.gx, fp.gy, fp.sx0, fp.sx1, fp.sy0,
                       fp.sy1, sat);
    return 0;
}

void launch_round_filter(const SplatList &list, const uint32_t *v_count, uint32_t n, const FramePlan *plan,
                         const uint16_t *sat, const uint32_t *tile_done, const FrameParams &fp, uint32_t *key_out,
                         uint32_t *dims_out, uint32_t *emit_sums, hipStream_t s) {
    if (n == 0) return;
    const uint32_t num_blocks = (n + PROJ_BLOCK - 1) / PROJ_BLOCK;
    hipLaunchKernelGGL(round_filter_kernel, dim3(std::min<uint32_t>(num_blocks, 1024u)), dim3(PROJ_BLOCK), 0, s, list,
                       v_count, plan, sat, tile_done, fp.gx, fp.gy, key_out, dims_out, emit_sums, num_blocks);
}

void launch_scan_blocks(const uint32_t *emit_sums, const uint4 *proj_sums, uint32_t num_blocks, uint64_t *block_base,
                        uint64_t capacity, uint64_t *total_out, uint32_t *d_sorted, uint32_t *overflow,
                        uint32_t *visible_out, uint32_t *last_tile_out, uint2 *bounds, uint32_t bounds_entries,
                        uint32_t *big_count, uint32_t *host_hint, const uint32_t *dc_parts, uint32_t *pairs_hint,
                        uint32_t *last_tile_copy, uint32_t *long_count, uint32_t *big_seen, hipStream_t s,
                        uint32_t frame_blocks) {
    // tile_bounds is allocated in multiples of 2 entries: cleared 16 bytes at a time
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(num_blocks ? (num_blocks + 1023u) / 1024u : 1u), dim3(1024), 0, s,
                       emit_sums, proj_sums, num_blocks, block_base, capacity, total_out, d_sorted, overflow, visible_out,
                       last_tile_out, reinterpret_cast<uint4 *>(bounds), (bounds_entries + 1u) / 2u, big_count,
                       host_hint, dc_parts, pairs_hint, last_tile_copy, long_count, big_seen, frame_blocks);
}

void launch_emit(const SplatList &list, const uint32_t *v_count, uint32_t n, const FrameParams &fp,
                 const uint32_t *emit_sums, const uint64_t *block_base, uint64_t capacity, uint32_t *keys,
                 uint32_t *values, uint32_t *big_count, uint32_t *big_list, bool narrow_keys, hipStream_t s,
                 uint32_t split, bool list_bigs, uint32_t big_hint) {
    if (n == 0) return;
    const dim3 grid((n + PROJ_BLOCK - 1) / PROJ_BLOCK, split ? split : 1u), block(PROJ_BLOCK);
    const uint32_t lb = list_bigs ? 1u : 0u;
    const TileMap map = tile_map_of(fp);
    if (narrow_keys) {
        uint16_t *k16 = reinterpret_cast<uint16_t *>(keys);
        hipLaunchKernelGGL(emit_kernel<uint16_t>, grid, block, 0, s, list, v_count, map, emit_sums, block_base, capacity,
                           k16, values, big_count, big_list, lb);
        if (list_bigs)
            hipLaunchKernelGGL(emit_big_kernel<uint16_t>, dim3(EMIT_BIG_X, emit_big_rows(big_hint)), dim3(256), 0, s, list, map,
                               block_base, capacity, k16, values, big_count, big_list);
    } else {
        hipLaunchKernelGGL(emit_kernel<uint32_t>, grid, block, 0, s, list, v_count, map, emit_sums, block_base, capacity,
                           keys, values, big_count, big_list, lb);
        if (list_bigs)
            hipLaunchKernelGGL(emit_big_kernel<uint32_t>, dim3(EMIT_BIG_X, emit_big_rows(big_hint)), dim3(256), 0, s, list, map,
                               block_base, capacity, keys, values, big_count, big_list);
    }
}

void launch_widen_keys(const uint16_t *keys16, const uint32_t *values, const uint32_t *splat_keys,
                       const uint32_t *d_count, uint32_t *keys_out, const TileMap &map, hipStream_t s) {
    hipLaunchKernelGGL(widen_keys_kernel, dim3(2048), dim3(256), 0, s, keys16, values, splat_keys, d_count, keys_out, map);
}

uint32_t emit_big_list_entries(uint64_t capacity) { return (uint32_t)(capacity / EMIT_BIG) + 2u; }

void launch_tile_counts(const uint32_t *dims, uint32_t *counts, uint32_t n, const uint32_t *block_skip, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(tile_counts_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, dims, counts, n, block_skip);
}

}  // namespace gsplat
