// Projection + key emission for gfx950 — replaces resources/shaders/compute/gsplat_projection.glsl.
//
// One lane per splat, 512-lane workgroups (8 wave64).  The scene is SoA (SceneSoA) so every load
// instruction of a wave is one contiguous 1 KiB run; culled splats touch 16 B.  The reference reserves
// key slots with a global atomicAdd (gsplat_projection.glsl:196), which makes the order of equal keys
// non-deterministic; here slots are the exclusive prefix sum of num_tiles_touched over ascending splat
// id: workgroup-local scan in this kernel (wave shuffles + LDS), a small scan of the workgroup totals,
// then emit_kernel writes (tile<<16 | depth16, id) pairs y-outer/x-inner (gsplat_projection.glsl:218-226).
// The SH colour (get_color, :94-121) is evaluated here only in "eager" frames; in "lazy" frames the
// compositor evaluates it for the splats it stages (sh_eval.h, raster.hip; api.hip chooses per frame).
//
// Arithmetic follows the contract in DESIGN.md §3 (compile with -ffp-contract=off): IEEE binary32,
// left-to-right sums, correctly rounded / and sqrt, pow(x,0.2) as a binary64 fifth root.
#include "gsplat_internal.h"
#include "sh_eval.h"

namespace gsplat {

namespace {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ float ease_out_cubic(float x) {  // gsplat_projection.glsl:87-90
    const float a = 1.0f - x;
    return 1.0f - (a * a) * a;
}

// pow(x, 0.2), gsplat_projection.glsl:190 — fifth root by 5 Newton steps in binary64.
__device__ __forceinline__ float pow02(float xf) {
    if (!(xf > 0.0f)) return 0.0f;
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

// Everything gsplat_projection.glsl:150-206 does for one splat: cull, project, colour, write RasterizeData.
// Returns num_tiles_touched (0 = the splat emits nothing); rect = packed tile rectangle (x0 | y0<<16, x1 | y1<<16),
// depth16 = the key's low half, last_plus1 = last tile of the unclamped rectangle + 1.
template <int EAGER>
__device__ __forceinline__ uint32_t project_splat(const SceneSoA &scene, uint32_t n, const FrameParams &fp, uint32_t id,
                                                  float4 *__restrict__ culled, uint2 &rect, uint32_t &depth_out,
                                                  uint32_t &last_plus1_out) {
    const float *V = fp.V, *P = fp.P;

    uint32_t count = 0, last_plus1 = 0;
    uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0, depth16 = 0;
    float ipx = 0, ipy = 0, px = 0, py = 0, pz = 0, opacity = 0, ca = 0, cb = 0, cc = 0, det = 1.0f;

    if (id < n) {
        const float4 pt = scene.pos_time[id];
        const float ms = fp.model_scale;
        // :160-166 frustum culling
        px = pt.x * ms; py = pt.y * ms; pz = pt.z * ms;
        const float vx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
        const float vy = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
        const float vz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
        const float vw = ((V[3] * px + V[7] * py) + V[11] * pz) + V[15];
        const float cx = ((P[0] * vx + P[4] * vy) + P[8] * vz) + P[12] * vw;
        const float cy = ((P[1] * vx + P[5] * vy) + P[9] * vz) + P[13] * vw;
        const float cz = ((P[2] * vx + P[6] * vy) + P[10] * vz) + P[14] * vw;
        const float cw = ((P[3] * vx + P[7] * vy) + P[11] * vz) + P[15] * vw;
        const float vb = cw * 1.2f;
        const bool culled_out = (cx < -vb) || (cy < -vb) || (cz < 0.0f) || (cx > vb) || (cy > vb) || (cz > cw);
        if (!culled_out) {
            const float4 A = scene.cov_a[id];
            const float4 Bc = scene.cov_b[id];
            // :169-174 load animation
            const float st = fp.time - pt.w;
            const float tf = ease_out_cubic(clampf(st, 0.0f, 1.0f));
            const float tfl = ease_out_cubic(clampf(st - 0.35f, 0.0f, 1.0f));
            opacity = (Bc.z * tfl) * tfl;
            const float smod = ms * (2.0f * (1.0f - tfl) + 1.0f * tfl);
            // :124-142 project_covariance
            const float C00 = (A.x * smod) * smod, C01 = (A.y * smod) * smod, C02 = (A.z * smod) * smod;
            const float C11 = (A.w * smod) * smod, C12 = (Bc.x * smod) * smod, C22 = (Bc.y * smod) * smod;
            const float tix = P[0], tiy = P[5];
            float fx = (fp.Wf * 0.5f) * tix, fy = (fp.Hf * 0.5f) * tiy;
            const float tfx = 1.0f / tix, tfy = 1.0f / tiy;
            const float zinv = 1.0f / vz;
            fx = fx * zinv;
            fy = fy * zinv;
            const float mx = clampf(vx * zinv, (-tfx) * 1.3f, tfx * 1.3f);
            const float my = clampf(vy * zinv, (-tfy) * 1.3f, tfy * 1.3f);
            const float j20 = (-fy) * mx;  // :135 focal.y in the x row
            const float j21 = (-fy) * my;
            float b0[3], b1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                b0[i] = V[i * 4 + 0] * fx + V[i * 4 + 2] * j20;
                b1[i] = V[i * 4 + 1] * fy + V[i * 4 + 2] * j21;
            }
            const float T00 = (b0[0] * C00 + b0[1] * C01) + b0[2] * C02;
            const float T01 = (b0[0] * C01 + b0[1] * C11) + b0[2] * C12;
            const float T02 = (b0[0] * C02 + b0[1] * C12) + b0[2] * C22;
            const float T10 = (b1[0] * C00 + b1[1] * C01) + b1[2] * C02;
            const float T11 = (b1[0] * C01 + b1[1] * C11) + b1[2] * C12;
            const float T12 = (b1[0] * C02 + b1[1] * C12) + b1[2] * C22;
            ca = ((T00 * b0[0] + T01 * b0[1]) + T02 * b0[2]) + 0.3f;
            cb = (T10 * b0[0] + T11 * b0[1]) + T12 * b0[2];
            cc = ((T10 * b1[0] + T11 * b1[1]) + T12 * b1[2]) + 0.3f;
            // :177-182
            det = ca * cc - cb * cb;
            const float mid = 0.5f * (ca + cc);
            const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
            const float l1 = mid + disc, l2 = mid - disc;
            if (det != 0.0f && !(l1 < 0.0f) && !(l2 < 0.0f)) {
                // :184-185
                const float nx = cx / cw, ny = cy / cw, nz = cz / cw;
                ipx = ((nx + 1.0f) * 0.5f - 1.0f * (1.0f - tf)) * fp.Wm1;
                ipy = ((ny + 1.0f) * 0.5f - 0.75f * (1.0f - tf)) * fp.Hm1;
                // :190-194, get_rect :144-148
                const float radius = (pow02(opacity) * 2.5f) * sqrtf(fmaxf(l1, l2));
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

    if (count) {
        // :202-206 RasterizeData.  The colour (:198-201, get_color) is NOT evaluated here: the compositor evaluates it
        // when it stages the splat (raster.hip), so the 12..192 bytes of SH coefficients are read only for splats that
        // are composited — at 6 M splats / deg 3 half of the visible splats never are (block early exit), and the SH
        // planes were 60 % of this kernel's traffic.  rgb slots are written as zeros (the parity tap fills them).
        // EAGER >= 0: this frame evaluates the colours here, for every visible splat, streaming the plane-major
        // coefficients (the better choice when most visible splats end up composited, api.hip picks per frame)
        float rgb[3] = {0.0f, 0.0f, 0.0f};
        if (EAGER >= 0) sh_color<(EAGER >= 0 ? EAGER : 0)>(scene.sh_planes + id, (size_t)n, px, py, pz, fp.cam, rgb);
        float4 *out = culled + (size_t)id * 3;
        out[0] = make_float4(ipx, ipy, px, py);                    // image_pos, pos_xy
        out[1] = make_float4(cc / det, (-cb) / det, ca / det, pz); // conic, pos_z
        out[2] = make_float4(rgb[0], rgb[1], rgb[2], opacity);     // color (rgb deferred when EAGER < 0), opacity
    }
    rect = make_uint2(x0 | (y0 << 16), x1 | (y1 << 16));
    depth_out = depth16;
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

// ---------------------------------------------------------------------------------------------------
// Split variant (GSPLAT_PROJECT=split): projection -> scan of workgroup totals -> emit, three kernels.
// ---------------------------------------------------------------------------------------------------
template <int EAGER>
__global__ __launch_bounds__(PROJ_BLOCK) void project_kernel(SceneSoA scene, uint32_t n, FrameParams fp,
                                                             float4 *__restrict__ culled,
                                                             uint32_t *__restrict__ local_off,
                                                             uint32_t *__restrict__ counts,
                                                             uint2 *__restrict__ rects,
                                                             uint32_t *__restrict__ depths,
                                                             uint4 *__restrict__ block_sums,
                                                             const uint32_t *__restrict__ block_skip) {
    __shared__ uint32_t wave_tot[PROJ_BLOCK / 64];
    __shared__ uint32_t wave_vis[PROJ_BLOCK / 64];
    __shared__ uint32_t wave_last[PROJ_BLOCK / 64];
    const uint32_t id = blockIdx.x * PROJ_BLOCK + threadIdx.x;
    if (block_skip != nullptr && block_skip[blockIdx.x]) {  // workgroup-uniform (block_cull_kernel)
        if (id < n) counts[id] = 0u;  // emit_kernel skips the workgroup (pairs == 0); the counts tap stays exact
        if (threadIdx.x == 0) block_sums[blockIdx.x] = make_uint4(0u, 0u, 0u, 1u);  // .w: skipped (debug tap)
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2 rect = make_uint2(0u, 0u);
    uint32_t depth16 = 0, last_plus1 = 0;
    const uint32_t count = project_splat<EAGER>(scene, n, fp, id, culled, rect, depth16, last_plus1);
    if (count) {
        rects[id] = rect;
        depths[id] = depth16;
    }

    // workgroup-local exclusive scan of count (deterministic stand-in for the atomicAdd of :196)
    // (no global atomics here: ~10^5 waves hitting one counter serialise at ~11 ns each — the per-workgroup
    // visible count and last tile ride along with the workgroup total and are reduced by scan_blocks_kernel)
    const uint32_t incl = wave_inclusive_scan(count, lane);
    const unsigned long long vis = __ballot(count != 0);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) last_plus1 = max(last_plus1, (uint32_t)__shfl_xor((int)last_plus1, d, 64));
    if (lane == 63) wave_tot[wave] = incl;
    if (lane == 0) {
        wave_vis[wave] = (uint32_t)__popcll(vis);
        wave_last[wave] = last_plus1;
    }
    __syncthreads();
    uint32_t wave_base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < PROJ_BLOCK / 64; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) wave_base += t;
        total += t;
    }
    if (id < n) {
        counts[id] = count;
        local_off[id] = wave_base + incl - count;
    }
    if (threadIdx.x == 0) {
        uint32_t v = 0, l = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) {
            v += wave_vis[w];
            l = max(l, wave_last[w]);
        }
        block_sums[blockIdx.x] = make_uint4(total, v, l, 0u);
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused variant (default): projection AND key emission in one kernel, no per-splat hand-off arrays.
// A workgroup draws a ticket = a chunk of 1024 consecutive splats (4 sub-tiles of 256), projects them, scans the tile
// counts (sub-tile by sub-tile, so slot order stays ascending splat id), publishes the chunk total and obtains the
// number of pairs emitted by all earlier chunks through decoupled look-back: ONE 64-bit granule per chunk
// {flag:2 | pairs:62}, relaxed agent-scope atomic store / loads (cdna_hip_programming.md G16 form R2: the datum is
// the flag).  Wave 0 inspects 64 predecessors per step (ballot for the nearest inclusive prefix).  Tickets make every
// predecessor a running workgroup (forward progress without residency assumptions); one ticket per 1024 splats keeps
// the ticket word far below its ~88 atomics/us saturation.  The chunk holding the last ticket writes D.
// ---------------------------------------------------------------------------------------------------
constexpr int CHUNK_TILES = 4;
constexpr uint32_t CHUNK = PROJ_BLOCK * CHUNK_TILES;
constexpr unsigned long long LB_AGG = 1ull << 62, LB_INC = 2ull << 62, LB_VALUE = (1ull << 62) - 1ull;
constexpr uint32_t LB_SPIN_LIMIT = 1u << 22;

template <int EAGER>
__global__ __launch_bounds__(PROJ_BLOCK) void project_emit_kernel(SceneSoA scene, uint32_t n, FrameParams fp,
                                                                  float4 *__restrict__ culled,
                                                                  uint32_t *__restrict__ counts,
                                                                  unsigned long long *chunk_status, uint32_t *ticket,
                                                                  uint2 *__restrict__ chunk_info, uint64_t capacity,
                                                                  uint32_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ values,
                                                                  uint64_t *__restrict__ total_out,
                                                                  uint32_t *__restrict__ d_sorted,
                                                                  uint32_t *__restrict__ overflow,
                                                                  uint32_t *__restrict__ error_flag) {
    // per-splat hand-off between the projection and emission halves lives in LDS (16 KiB), not in HBM
    __shared__ uint2 s_rect[CHUNK_TILES][PROJ_BLOCK];     // packed tile rectangle (empty = emits nothing)
    __shared__ uint32_t s_depth[CHUNK_TILES][PROJ_BLOCK];
    __shared__ uint32_t s_excl[CHUNK_TILES][PROJ_BLOCK];  // slot offset of the splat within the chunk
    __shared__ uint32_t wave_tot[PROJ_BLOCK / 64];
    __shared__ uint32_t s_ticket;
    __shared__ unsigned long long s_base;
    __shared__ uint32_t red_vis[PROJ_BLOCK / 64], red_last[PROJ_BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t num_chunks = (n + CHUNK - 1) / CHUNK;

    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t chunk = s_ticket;
    if (chunk >= num_chunks) return;
    const uint32_t first = chunk * CHUNK;

    // ---- project the chunk's splats, sub-tile by sub-tile; tile counts -> exclusive offsets within the chunk
    uint32_t tile_base = 0;  // pairs of the previous sub-tiles of this chunk
    uint32_t my_vis = 0, my_last = 0;
#pragma unroll 1
    for (int t = 0; t < CHUNK_TILES; ++t) {
        const uint32_t id = first + t * PROJ_BLOCK + threadIdx.x;
        uint2 rect = make_uint2(0u, 0u);
        uint32_t depth16 = 0, last_plus1 = 0;
        const uint32_t count = project_splat<EAGER>(scene, n, fp, id, culled, rect, depth16, last_plus1);
        s_rect[t][threadIdx.x] = count ? rect : make_uint2(0u, 0u);
        s_depth[t][threadIdx.x] = depth16;
        if (id < n) counts[id] = count;
        my_vis += count != 0;
        my_last = max(my_last, last_plus1);
        const uint32_t incl = wave_inclusive_scan(count, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) {
            const uint32_t v = wave_tot[w];
            if (w < wave) wave_base += v;
            total += v;
        }
        s_excl[t][threadIdx.x] = tile_base + wave_base + incl - count;
        tile_base += total;
        __syncthreads();
    }
    const uint32_t chunk_total = tile_base;

    // ---- publish, look back (wave 0), broadcast the chunk's global base
    if (wave == 0) {
        if (lane == 0)
            __hip_atomic_store(chunk_status + chunk, (unsigned long long)chunk_total | (chunk == 0 ? LB_INC : LB_AGG),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long base = 0;
        if (chunk > 0) {
            int64_t hi = (int64_t)chunk - 1;  // nearest predecessor not yet accounted for
            uint32_t spins = 0;
            for (;;) {
                const int64_t q = hi - lane;
                const unsigned long long w = q >= 0 ? __hip_atomic_load(chunk_status + q, __ATOMIC_RELAXED,
                                                                        __HIP_MEMORY_SCOPE_AGENT)
                                                    : LB_INC;  // before chunk 0: empty inclusive prefix
                const unsigned long long flag = w >> 62;
                const unsigned long long not_ready = __ballot(flag == 0);
                const unsigned long long inc = __ballot(flag == 2);
                const int first_nr = not_ready ? __builtin_ctzll(not_ready) : 64;
                const int first_inc = inc ? __builtin_ctzll(inc) : 64;
                const int take = first_inc < first_nr ? first_inc + 1 : first_nr;  // lanes [0, take) are usable
                unsigned long long v = lane < take ? (w & LB_VALUE) : 0ull;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                base += v;
                if (first_inc < first_nr) break;
                hi -= take;
                if (take == 0) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > LB_SPIN_LIMIT) {
                        if (lane == 0) *error_flag = 1u;
                        break;
                    }
                } else {
                    spins = 0;
                }
            }
            if (lane == 0)
                __hip_atomic_store(chunk_status + chunk, (base + chunk_total) | LB_INC, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            s_base = base;
            if (chunk == num_chunks - 1) {  // the last chunk knows D
                const unsigned long long total = base + chunk_total;
                *total_out = total;
                *d_sorted = (uint32_t)(total < capacity ? total : capacity);
                *overflow = total > capacity ? 1u : 0u;
            }
        }
    }
    // per-chunk visible count / last tile (reduced by reduce_chunks_kernel)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        my_vis += __shfl_xor(my_vis, d, 64);
        my_last = max(my_last, (uint32_t)__shfl_xor((int)my_last, d, 64));
    }
    if (lane == 0) { red_vis[wave] = my_vis; red_last[wave] = my_last; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = 0, l = 0;
#pragma unroll
        for (int w = 0; w < PROJ_BLOCK / 64; ++w) { v += red_vis[w]; l = max(l, red_last[w]); }
        chunk_info[chunk] = make_uint2(v, l);
    }
    const unsigned long long base = s_base;

    // ---- emit (gsplat_projection.glsl:218-226), y outer / x inner, slots in ascending splat id
#pragma unroll
    for (int t = 0; t < CHUNK_TILES; ++t) {
        const uint2 r = s_rect[t][threadIdx.x];
        const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, x1 = r.y & 0xFFFFu, y1 = r.y >> 16;
        if (x1 <= x0 || y1 <= y0) continue;
        const uint32_t id = first + t * PROJ_BLOCK + threadIdx.x;
        const uint32_t depth = s_depth[t][threadIdx.x];
        unsigned long long off = base + s_excl[t][threadIdx.x];
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = x0; x < x1; ++x) {
                if (off < capacity) {  // never write past the key budget
                    keys[off] = ((y * fp.gx + x) << 16) | depth;
                    values[off] = id;
                }
                ++off;
            }
    }
}

// visible count and the frame's last tile from the per-chunk records (one workgroup)
__global__ __launch_bounds__(1024) void reduce_chunks_kernel(const uint2 *__restrict__ chunk_info, uint32_t num_chunks,
                                                             uint32_t *__restrict__ visible_out,
                                                             uint32_t *__restrict__ last_tile_out) {
    __shared__ uint32_t vis_s[16], last_s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = 0, l = 0;
    for (uint32_t i = threadIdx.x; i < num_chunks; i += 1024) {
        const uint2 c = chunk_info[i];
        v += c.x;
        l = max(l, c.y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d, 64);
        l = max(l, (uint32_t)__shfl_xor((int)l, d, 64));
    }
    if (lane == 0) { vis_s[wave] = v; last_s[wave] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t vv = 0, ll = 0;
        for (int w = 0; w < 16; ++w) { vv += vis_s[w]; ll = max(ll, last_s[w]); }
        *visible_out = vv;
        *last_tile_out = ll;
    }
}

// Exclusive scan of the workgroup totals (N/512 entries); 64-bit bases so a pathological D cannot wrap.  Also reduces
// the visible count and the frame's last tile, finalises D / min(D, capacity) / overflow and clears tile_bounds.
// One workgroup per 1024 workgroup totals, no inter-workgroup dependency: workgroup k first reduces ALL totals before
// its slice (k x 16 KiB of reads — 1 MiB over the whole grid at 6 M splats), then scans its own 1024.  Two memory
// round trips instead of a serial loop in one workgroup (28 us -> a few us; the serial form was 17 % of a rank's
// projection pass in an 8-way stripe shard).  The last workgroup sees every total and writes the frame counters.
__global__ __launch_bounds__(1024) void scan_blocks_kernel(const uint4 *__restrict__ block_sums,
                                                           uint32_t num_blocks, uint64_t *__restrict__ block_base,
                                                           uint64_t capacity, uint64_t *__restrict__ total_out,
                                                           uint32_t *__restrict__ d_sorted,
                                                           uint32_t *__restrict__ overflow,
                                                           uint32_t *__restrict__ visible_out,
                                                           uint32_t *__restrict__ last_tile_out,
                                                           uint4 *__restrict__ bounds_as_uint4, uint32_t bounds_uint4s,
                                                           uint32_t *__restrict__ big_count,
                                                           const uint32_t *__restrict__ tile_staged, uint32_t num_tiles,
                                                           uint32_t *__restrict__ host_hint) {
    __shared__ uint64_t wave_pre[16], wave_own[16];
    __shared__ uint32_t vis_s[16], last_s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // gaussian_splatting_rasterizer.gd:128 buffer_clear(tile_bounds): done here (a few KiB..260 KiB) instead of a
    // separate fill launch; boundaries_kernel runs after the whole sort, long after this
    for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < bounds_uint4s; i += gridDim.x * 1024u)
        bounds_as_uint4[i] = make_uint4(0u, 0u, 0u, 0u);

    const uint32_t first = blockIdx.x * 1024u;
    uint64_t pre = 0;  // pairs of the workgroups before this slice
    uint32_t vis = 0, last = 0;
    for (uint32_t i = threadIdx.x; i < first; i += 1024u) {
        const uint4 bs = block_sums[i];
        pre += bs.x;
        vis += bs.y;
        last = max(last, bs.z);
    }
    const uint32_t i = first + threadIdx.x;
    const uint4 own = i < num_blocks ? block_sums[i] : make_uint4(0u, 0u, 0u, 0u);
    vis += own.y;
    last = max(last, own.z);
    uint64_t incl = own.x;
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
    if (i < num_blocks) block_base[i] = base + incl - own.x;
    // the last workgroup also adds up what the compositor staged per tile in the PREVIOUS frame (D_c) and posts it,
    // with this frame's visible count, to host-mapped memory: the host picks the next frame's colour mode from them
    uint32_t dc_prev = 0;
    if (blockIdx.x == gridDim.x - 1 && host_hint != nullptr) {
        __shared__ uint32_t dc_s[16];
        for (uint32_t t = threadIdx.x; t < num_tiles; t += 1024u) dc_prev += tile_staged[t];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) dc_prev += __shfl_xor(dc_prev, d, 64);
        if (lane == 0) dc_s[wave] = dc_prev;
        __syncthreads();
        dc_prev = 0;
        for (int w = 0; w < 16; ++w) dc_prev += dc_s[w];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint64_t total = own_total;
        uint32_t vv = 0, l = 0;
        for (int w = 0; w < 16; ++w) { total += wave_pre[w]; vv += vis_s[w]; l = max(l, last_s[w]); }
        *total_out = total;
        *d_sorted = (uint32_t)(total < capacity ? total : capacity);
        *overflow = total > capacity ? 1u : 0u;
        *visible_out = vv;
        *last_tile_out = l;
        if (host_hint != nullptr) {
            host_hint[0] = vv;
            host_hint[1] = dc_prev;
            host_hint[2] = ++big_count[2];  // frames posted so far, counted in device memory (third word of the block)
        }
        big_count[0] = 0u;  // emit_kernel's list of big rectangles starts empty
        big_count[1] = 0u;  // ... and the tile sort's list of long segments (the next word of the counter block)
    }
}

// gsplat_projection.glsl:218-226: duplicate (key, id) over the tile rectangle, y outer / x inner.
// The slot of every pair is fixed by the scans (block_base + local_off + y-outer/x-inner index), so any distribution of
// the writes gives the same buffers; two levels keep the stores coalesced and the load balanced whatever the splat sizes:
//  * emit_kernel — wave-cooperative: the pairs of a wave's 64 splats are numbered 0..T-1 (wave scan), lane l writes
//    pairs l, l+64, ... and finds the owning splat by a 6-step binary search over the lanes' inclusive ends (shuffles).
//  * splats covering more than EMIT_BIG tiles are only *listed* there and written by emit_big_kernel, where the whole
//    grid shares each rectangle.  (A per-lane loop over its own rectangle made 2000 screen-filling splats cost 1.2 ms,
//    and any per-wave scheme still serialises when such splats sit next to each other in id order — which is what a
//    Morton-ordered scene does with the region next to the camera.  tools/big_splats.py is the stress case.)
// (A no-wait look-back over block_sums inside this kernel was tried instead of scan_blocks_kernel: with ~2000
// workgroups in flight nobody has published a prefix nearby, every workgroup walks ~2000 entries, 2.5x slower.)
constexpr uint32_t EMIT_BIG = 512;

__device__ __forceinline__ void write_pair(uint32_t j, uint32_t x0, uint32_t y0, uint32_t wx, uint32_t depth, uint32_t id,
                                           uint32_t gx, uint64_t off, uint64_t capacity, uint32_t *__restrict__ keys,
                                           uint32_t *__restrict__ values) {
    // j / wx without an integer divide: float estimate (j < 2^24), corrected by at most one
    uint32_t q = (uint32_t)((float)j * (1.0f / (float)wx));
    int32_t rem = (int32_t)(j - q * wx);
    if (rem < 0) { --q; rem += (int32_t)wx; }
    if (rem >= (int32_t)wx) { ++q; rem -= (int32_t)wx; }
    if (off < capacity) {  // never write past the key budget
        keys[off] = (((y0 + q) * gx + (x0 + (uint32_t)rem)) << 16) | depth;
        values[off] = id;
    }
}

__global__ __launch_bounds__(PROJ_BLOCK) void emit_kernel(uint32_t n, uint32_t gx,
                                                          const uint32_t *__restrict__ local_off,
                                                          const uint32_t *__restrict__ counts,
                                                          const uint2 *__restrict__ rects,
                                                          const uint32_t *__restrict__ depths,
                                                          const uint4 *__restrict__ block_sums,
                                                          const uint64_t *__restrict__ block_base, uint64_t capacity,
                                                          uint32_t *__restrict__ keys, uint32_t *__restrict__ values,
                                                          uint32_t *__restrict__ big_count,
                                                          uint32_t *__restrict__ big_list) {
    // a workgroup whose 512 splats emit nothing (most of them in a tile-stripe shard of a Morton-ordered scene)
    // leaves after one 16-byte read
    if (block_sums[blockIdx.x].x == 0u) return;
    const uint32_t id = blockIdx.x * PROJ_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t count = id < n ? counts[id] : 0u;
    uint32_t excl = 0, depth = 0;
    uint2 r = make_uint2(0u, 0u);
    if (count) {
        excl = local_off[id];  // offset within the workgroup's range (ascending with id)
        r = rects[id];
        depth = depths[id];
    }
    const uint64_t base = block_base[blockIdx.x];

    // big rectangles: wave-aggregated append to the list (order in the list is irrelevant, slots are fixed)
    const bool big = count > EMIT_BIG && base + excl < capacity;
    const unsigned long long big_mask = __ballot(big);
    if (big_mask) {
        uint32_t first_slot = 0;
        if (lane == (int)__builtin_ctzll(big_mask)) first_slot = atomicAdd(big_count, (uint32_t)__popcll(big_mask));
        first_slot = __shfl(first_slot, (int)__builtin_ctzll(big_mask), 64);
        if (big) big_list[first_slot + (uint32_t)__popcll(big_mask & ((1ull << lane) - 1ull))] = id;
    }

    const uint32_t small = count > EMIT_BIG ? 0u : count;
    const uint32_t incl = wave_inclusive_scan(small, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0u) return;  // wave-uniform
    const uint32_t pair0 = incl - small;  // number of this lane's first pair within the wave
    const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, wx = (r.y & 0xFFFFu) - x0;
    const uint32_t wave_id0 = blockIdx.x * PROJ_BLOCK + (threadIdx.x & ~63u);
    for (uint32_t p = (uint32_t)lane; p < ((total + 63u) & ~63u); p += 64u) {
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
        const uint32_t s_wx = __shfl(wx, src, 64), s_depth = __shfl(depth, src, 64);
        if (p < total) {
            const uint32_t j = p - s_pair0;  // index inside the splat's rectangle, y outer / x inner
            write_pair(j, s_x0, s_y0, s_wx, s_depth, wave_id0 + (uint32_t)src, gx, base + s_excl + j, capacity, keys,
                       values);
        }
    }
}

// grid (EMIT_BIG_X, EMIT_BIG_Y): blockIdx.y strides over the listed splats, blockIdx.x over 256-pair pieces of one
constexpr uint32_t EMIT_BIG_X = 8, EMIT_BIG_Y = 128;
__global__ __launch_bounds__(256) void emit_big_kernel(uint32_t gx, const uint32_t *__restrict__ local_off,
                                                       const uint32_t *__restrict__ counts,
                                                       const uint2 *__restrict__ rects,
                                                       const uint32_t *__restrict__ depths,
                                                       const uint64_t *__restrict__ block_base, uint64_t capacity,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ values,
                                                       const uint32_t *__restrict__ big_count,
                                                       const uint32_t *__restrict__ big_list) {
    const uint32_t nb = *big_count;
    for (uint32_t e = blockIdx.y; e < nb; e += gridDim.y) {
        const uint32_t id = big_list[e];
        const uint32_t count = counts[id], depth = depths[id];
        const uint2 r = rects[id];
        const uint64_t off0 = block_base[id / PROJ_BLOCK] + local_off[id];
        const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, wx = (r.y & 0xFFFFu) - x0;
        for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < count; j += gridDim.x * 256u)
            write_pair(j, x0, y0, wx, depth, id, gx, off0 + j, capacity, keys, values);
    }
}

}  // namespace

void launch_project(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                    uint32_t *local_off, uint32_t *counts, uint2 *rects, uint32_t *depths, uint4 *block_sums,
                    const float4 *block_bounds, uint32_t *block_skip, hipStream_t s) {
    if (n == 0) return;
    const dim3 grid((n + PROJ_BLOCK - 1) / PROJ_BLOCK), block(PROJ_BLOCK);
    const bool cull = fp.cull_mode != 0u && block_bounds != nullptr && block_skip != nullptr;
    if (cull)
        hipLaunchKernelGGL(block_cull_kernel, dim3((grid.x + 255u) / 256u), dim3(256), 0, s, fp, block_bounds, grid.x,
                           block_skip);
    const uint32_t *skip = cull ? block_skip : nullptr;
#define GSPLAT_LAUNCH_P(E)                                                                                         \
    hipLaunchKernelGGL(project_kernel<E>, grid, block, 0, s, scene, n, fp, culled, local_off, counts, rects, depths, \
                       block_sums, skip)
    switch (sh_degree) {  // -1: colours left to the compositor
        case 0: GSPLAT_LAUNCH_P(0); break;
        case 1: GSPLAT_LAUNCH_P(1); break;
        case 2: GSPLAT_LAUNCH_P(2); break;
        case 3: GSPLAT_LAUNCH_P(3); break;
        default: GSPLAT_LAUNCH_P(-1); break;
    }
#undef GSPLAT_LAUNCH_P
}

void launch_block_bounds(const SceneSoA &scene, uint32_t n, float4 *block_bounds, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(block_bounds_kernel, dim3((n + PROJ_BLOCK - 1) / PROJ_BLOCK), dim3(PROJ_BLOCK), 0, s, scene, n,
                       block_bounds);
}

void launch_project_emit(const SceneSoA &scene, uint32_t n, const FrameParams &fp, int sh_degree, float4 *culled,
                         uint32_t *counts, unsigned long long *chunk_status, uint32_t *ticket, uint2 *chunk_info,
                         uint64_t capacity, uint32_t *keys, uint32_t *values, uint64_t *total_out, uint32_t *d_sorted,
                         uint32_t *overflow, uint32_t *visible_out, uint32_t *last_tile_out, uint32_t *error_flag,
                         hipStream_t s) {
    const uint32_t num_chunks = (n + CHUNK - 1) / CHUNK;
    (void)hipMemsetAsync(chunk_status, 0, (size_t)(num_chunks ? num_chunks : 1) * sizeof(unsigned long long), s);
    (void)hipMemsetAsync(ticket, 0, sizeof(uint32_t), s);
    if (n == 0) {
        (void)hipMemsetAsync(total_out, 0, sizeof(uint64_t), s);
        return;
    }
    const dim3 grid(num_chunks), block(PROJ_BLOCK);
#define GSPLAT_LAUNCH_PE(E)                                                                                        \
    hipLaunchKernelGGL(project_emit_kernel<E>, grid, block, 0, s, scene, n, fp, culled, counts, chunk_status, ticket, \
                       chunk_info, capacity, keys, values, total_out, d_sorted, overflow, error_flag)
    switch (sh_degree) {
        case 0: GSPLAT_LAUNCH_PE(0); break;
        case 1: GSPLAT_LAUNCH_PE(1); break;
        case 2: GSPLAT_LAUNCH_PE(2); break;
        case 3: GSPLAT_LAUNCH_PE(3); break;
        default: GSPLAT_LAUNCH_PE(-1); break;
    }
#undef GSPLAT_LAUNCH_PE
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3(1), dim3(1024), 0, s, chunk_info, num_chunks, visible_out,
                       last_tile_out);
}

uint32_t project_num_chunks(uint32_t n) { return (n + CHUNK - 1) / CHUNK; }

void launch_scan_blocks(const uint4 *block_sums, uint32_t num_blocks, uint64_t *block_base, uint64_t capacity,
                        uint64_t *total_out, uint32_t *d_sorted, uint32_t *overflow, uint32_t *visible_out,
                        uint32_t *last_tile_out, uint2 *bounds, uint32_t bounds_entries, uint32_t *big_count,
                        const uint32_t *tile_staged, uint32_t num_tiles, uint32_t *host_hint, hipStream_t s) {
    // tile_bounds (+ the tile segments behind it) is allocated in multiples of 2 entries: cleared 16 bytes at a time
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(num_blocks ? (num_blocks + 1023u) / 1024u : 1u), dim3(1024), 0, s, block_sums, num_blocks, block_base, capacity,
                       total_out, d_sorted, overflow, visible_out, last_tile_out, reinterpret_cast<uint4 *>(bounds),
                       (bounds_entries + 1u) / 2u, big_count, tile_staged, num_tiles, host_hint);
}

void launch_emit(uint32_t n, const FrameParams &fp, const uint32_t *local_off, const uint32_t *counts,
                 const uint2 *rects, const uint32_t *depths, const uint4 *block_sums, const uint64_t *block_base,
                 uint64_t capacity, uint32_t *keys, uint32_t *values, uint32_t *big_count, uint32_t *big_list,
                 hipStream_t s) {
    if (n == 0) return;
    const dim3 grid((n + PROJ_BLOCK - 1) / PROJ_BLOCK), block(PROJ_BLOCK);
    hipLaunchKernelGGL(emit_kernel, grid, block, 0, s, n, fp.gx, local_off, counts, rects, depths, block_sums,
                       block_base, capacity, keys, values, big_count, big_list);
    hipLaunchKernelGGL(emit_big_kernel, dim3(EMIT_BIG_X, EMIT_BIG_Y), dim3(256), 0, s, fp.gx, local_off, counts, rects,
                       depths, block_base, capacity, keys, values, big_count, big_list);
}

uint32_t emit_big_list_entries(uint64_t capacity) { return (uint32_t)(capacity / EMIT_BIG) + 2u; }

}  // namespace gsplat
