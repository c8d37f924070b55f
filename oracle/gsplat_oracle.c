/*
 * gsplat_oracle.c — CPU restatement of the reference's forward Gaussian-splat pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under godotgaussiansplatting_amd/ may link, import or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it (as the checker /
 * the timed CPU baseline, never as the product).
 *
 * PARITY STATUS: "parity unpinned".  The reference (2Retr0/GodotGaussianSplatting @ 2024_10_08) ships
 * no tests, no golden vectors and its default scene (resources/demo.ply) is a missing blob; Godot,
 * Vulkan and a GLSL compiler are absent from this image, so the shaders cannot be executed.  This file
 * is pinned only by (a) following the shader text line by line (citations below, relative to
 * /root/reference), (b) hand-derived known-answer tests in tests/test_oracle_kat.py and (c) an
 * independent float64 NumPy twin (oracle/numpy_twin.py) that evaluates the literal GLSL expressions.
 *
 * ARITHMETIC CONTRACT.  GLSL leaves the rounding of exp/pow/normalize, FMA
 * contraction and matrix-product summation order to the driver, yet the pipeline is discontinuous in
 * those values (tile rectangles, 16-bit depth codes, the t > 1/255 stop rule, the block early-exit
 * sum).  To make "bit-exact tile indices" and "RGBA within 1e-4" testable, this restatement fixes one
 * member of the family of valid evaluations:
 *   - all arithmetic is IEEE-754 binary32 with round-to-nearest-even, NO implicit FMA contraction
 *     (compile with -ffp-contract=off); fmaf() appears only where the contract says "fma";
 *   - sums of products (matrix products, dot products) are accumulated left to right, ascending index;
 *   - divisions and sqrt are correctly rounded;
 *   - pow(x, 0.2) is the real fifth root evaluated in binary64 by 5 Newton steps from a bit-level
 *     initial guess, then rounded once to binary32 (gso_pow02);
 *   - exp(x) in the compositor is gso_exp2(x*log2(e)) = 2^n * p(f), x*log2(e) clamped to [-125,126], a degree-5 polynomial with
 *     p(0) == 1 exactly and <= 2.8 ulp error (inside Vulkan's 3+2|x| ulp allowance for exp);
 *   - the compositor's quadratic form is evaluated as  dx*(hx*dx + hy*dy) + (hz*dy)*dy  with
 *     (hx,hy,hz) = (-0.5*cx, -cy, -0.5*cz)*log2(e) prepared once per splat; `t *= 1-alpha` is
 *     evaluated as t - alpha*t.  These are algebraically the expressions of gsplat_render.glsl:84-90.
 * The HIP kernels implement the same contract, so integer outputs AND the image are bit-identical
 * to this file on the same inputs.
 *
 * Deterministic member of the reference's non-deterministic emission order: the slot
 * reservation of gsplat_projection.glsl:196 (a global atomicAdd) is replaced by the exclusive prefix
 * sum of num_tiles_touched over ascending splat id.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GSO_TILE 16
#define GSO_BLOCK 256 /* gsplat_render.glsl:9 WORKGROUP_SIZE */

/* gsplat_projection.glsl:6-21 (the GLSL literals are binary32 constants). */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2_0 = 1.0925484305920792f;
static const float SH_C2_1 = 1.0925484305920792f;
static const float SH_C2_2 = 0.31539156525252005f;
static const float SH_C2_3 = 1.0925484305920792f;
static const float SH_C2_4 = 0.5462742152960396f;
static const float SH_C3_0 = 0.5900435899266435f;
static const float SH_C3_1 = 2.890611442640554f;
static const float SH_C3_2 = 0.4570457994644658f;
static const float SH_C3_3 = 0.3731763325901154f;
static const float SH_C3_4 = 0.4570457994644658f;
static const float SH_C3_5 = 1.445305721320277f;
static const float SH_C3_6 = 0.5900435899266435f;

/* Per-frame inputs = the 32 B uniform block (gsplat_projection.glsl:75-80, written at
 * gaussian_splatting_rasterizer.gd:126), the 128 B push constant (gsplat_projection.glsl:82-85,
 * built at gaussian_splatting_rasterizer.gd:181-193) and the render push constant
 * (gsplat_render.glsl:40-43, gaussian_splatting_rasterizer.gd:158). */
typedef struct {
    float view[16]; /* column-major mat4 */
    float proj[16]; /* column-major mat4 */
    float cam_pos[3];
    float model_scale;
    int32_t width, height;
    float time;
    float heatmap_factor;
    uint32_t target_tile;
    /* stripe clamp for the multi-GPU shard (tile units, [x0,x1) x [y0,y1)); full grid = no clamp */
    uint32_t stripe_x0, stripe_x1, stripe_y0, stripe_y1;
} gso_frame;

typedef struct {
    uint64_t visible;     /* V: splats that wrote RasterizeData */
    uint64_t emitted;     /* D before the capacity clamp */
    uint64_t sorted;      /* min(D, capacity) */
    uint64_t composited;  /* D_c = sum over tiles of min(n, 256*iterations executed) */
    uint64_t evals;       /* splat-pixel evaluations actually performed */
    uint64_t wave_steps;  /* (wave64, splat) steps with at least one live pixel: what a wave64 machine executes */
    uint64_t wave_full;   /* ... of which at least one live pixel is above the exp cutoff (full evaluation) */
    int32_t overflow;
    int32_t sig_bits;
} gso_stats;

/* ------------------------------------------------------------------------------------------------
 * contract math
 * ---------------------------------------------------------------------------------------------- */

/* pow(x, 0.2) of gsplat_projection.glsl:190. */
float gso_pow02(float xf) {
    if (!(xf > 0.0f)) return 0.0f;
    double x = (double)xf;
    int64_t i;
    memcpy(&i, &x, 8);
    const int64_t B = 0x3FF0000000000000LL;
    i = i / 5 + (B - B / 5);
    double r;
    memcpy(&r, &i, 8);
    for (int k = 0; k < 5; ++k) {
        double r2 = r * r;
        double r4 = r2 * r2;
        r = (4.0 * r + x / r4) / 5.0;
    }
    return (float)r;
}

#define GSO_LOG2E 0x1.715476p+0f
/* exp(power) is taken as exactly 0 when power*log2(e) < -32 (it would be < 2.4e-10): the splat then changes neither
 * the colour nor the transmittance of that pixel.  A real exp() underflows to 0 as well, only later (-149); choosing
 * the threshold lets a wave64 skip a splat none of its pixels can see.  Bounded deviation, DESIGN.md §3 item 4. */
/* which wave64 of the 16x16 workgroup a pixel (local index p = y*16+x) belongs to — statistics only.
 * The kernels map a wave to an 8x8 pixel quadrant (compact footprint: more splats are out of reach of a whole wave). */
#ifndef GSO_WAVE_OF
#define GSO_WAVE_OF(p) ((((p) >> 4) >> 3) * 2 + (((p) & 15) >> 3))
#endif
#ifndef GSO_EXP_CUTOFF
#define GSO_EXP_CUTOFF (-32.0f)
#endif

/* 2^y, y clamped to [-125, 126] (the result is always a normal number: the kernels build it by adding n to the
 * exponent field of p). */
float gso_exp2(float y) {
    y = fminf(fmaxf(y, -125.0f), 126.0f);
    float n = rintf(y); /* round-half-even */
    float f = y - n;
    float q = fmaf(0x1.5bba18p-10f, f, 0x1.3cea88p-7f);
    q = fmaf(q, f, 0x1.c6b752p-5f);
    q = fmaf(q, f, 0x1.ebf9bcp-3f);
    q = fmaf(q, f, 0x1.62e42ap-1f);
    float p = fmaf(q, f, 1.0f);
    return ldexpf(p, (int)n);
}

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* gsplat_projection.glsl:87-90 */
static inline float ease_out_cubic(float x) {
    float a = 1.0f - x;
    return 1.0f - (a * a) * a;
}

typedef struct {
    float raster[12]; /* RasterizeData, gsplat_projection.glsl:42-48 */
    uint32_t x0, y0, x1, y1;
    uint32_t depth16;
    uint32_t count;
    uint32_t last_tile_plus1; /* last tile of the UNCLAMPED rectangle + 1 (0 = none) */
} gso_proj;

/* gsplat_projection.glsl:150-206 for one splat record (60 floats, gsplat_projection.glsl:33-40).
 * Returns 1 if the splat survives and fills *o, else 0. */
static int project_one(const float *s, const gso_frame *fr, uint32_t gx, uint32_t gy, gso_proj *o) {
    const float *V = fr->view, *P = fr->proj;
    const float ms = fr->model_scale;
    const float W = (float)fr->width, H = (float)fr->height;
    o->last_tile_plus1 = 0;
    o->count = 0;

    /* :160-166 frustum culling */
    const float px = s[0] * ms, py = s[1] * ms, pz = s[2] * ms;
    const float vx = ((V[0] * px + V[4] * py) + V[8] * pz) + V[12];
    const float vy = ((V[1] * px + V[5] * py) + V[9] * pz) + V[13];
    const float vz = ((V[2] * px + V[6] * py) + V[10] * pz) + V[14];
    const float vw = ((V[3] * px + V[7] * py) + V[11] * pz) + V[15];
    const float cx = ((P[0] * vx + P[4] * vy) + P[8] * vz) + P[12] * vw;
    const float cy = ((P[1] * vx + P[5] * vy) + P[9] * vz) + P[13] * vw;
    const float cz = ((P[2] * vx + P[6] * vy) + P[10] * vz) + P[14] * vw;
    const float cw = ((P[3] * vx + P[7] * vy) + P[11] * vz) + P[15] * vw;
    const float vb = cw * 1.2f;
    if (cx < -vb || cy < -vb || cz < 0.0f || cx > vb || cy > vb || cz > cw) return 0;

    /* :169-174 load animation */
    const float st = fr->time - s[3];
    const float tf = ease_out_cubic(clampf(st, 0.0f, 1.0f));
    const float tfl = ease_out_cubic(clampf(st - 0.35f, 0.0f, 1.0f));
    const float opacity = (s[10] * tfl) * tfl;
    const float smod = ms * (2.0f * (1.0f - tfl) + 1.0f * tfl); /* mix(2,1,tfl) */

    /* :124-142 project_covariance */
    float c3[6];
    for (int i = 0; i < 6; ++i) c3[i] = (s[4 + i] * smod) * smod;
    /* symmetric 3x3, :29 DECODE_COVARIANCE */
    const float C00 = c3[0], C01 = c3[1], C02 = c3[2], C11 = c3[3], C12 = c3[4], C22 = c3[5];
    const float tix = P[0], tiy = P[5];
    float fx = (W * 0.5f) * tix, fy = (H * 0.5f) * tiy;
    const float tfx = 1.0f / tix, tfy = 1.0f / tiy;
    const float zinv = 1.0f / vz;
    fx = fx * zinv;
    fy = fy * zinv;
    const float mx = clampf(vx * zinv, (-tfx) * 1.3f, tfx * 1.3f);
    const float my = clampf(vy * zinv, (-tfy) * 1.3f, tfy * 1.3f);
    const float j20 = (-fy) * mx; /* :135 uses focal.y in the x row */
    const float j21 = (-fy) * my;
    /* inv_view(i,k) = V[i*4+k] (transpose of the upper 3x3); b = inv_view * J, column 2 of J is 0 and
     * its literal-zero products are dropped. */
    float b0[3], b1[3];
    for (int i = 0; i < 3; ++i) {
        b0[i] = V[i * 4 + 0] * fx + V[i * 4 + 2] * j20;
        b1[i] = V[i * 4 + 1] * fy + V[i * 4 + 2] * j21;
    }
    /* T = b^T * C3 (rows 0,1), cov2 = T * b */
    const float T00 = (b0[0] * C00 + b0[1] * C01) + b0[2] * C02;
    const float T01 = (b0[0] * C01 + b0[1] * C11) + b0[2] * C12;
    const float T02 = (b0[0] * C02 + b0[1] * C12) + b0[2] * C22;
    const float T10 = (b1[0] * C00 + b1[1] * C01) + b1[2] * C02;
    const float T11 = (b1[0] * C01 + b1[1] * C11) + b1[2] * C12;
    const float T12 = (b1[0] * C02 + b1[1] * C12) + b1[2] * C22;
    const float ca = ((T00 * b0[0] + T01 * b0[1]) + T02 * b0[2]) + 0.3f;  /* cov_2d[0][0] + 0.3 */
    const float cb = (T10 * b0[0] + T11 * b0[1]) + T12 * b0[2];           /* cov_2d[0][1] */
    const float cc = ((T10 * b1[0] + T11 * b1[1]) + T12 * b1[2]) + 0.3f;  /* cov_2d[1][1] + 0.3 */

    /* :177-182 */
    const float det = ca * cc - cb * cb;
    if (det == 0.0f) return 0;
    const float mid = 0.5f * (ca + cc);
    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float l1 = mid + disc, l2 = mid - disc;
    if (l1 < 0.0f || l2 < 0.0f) return 0;

    /* :184-185 */
    const float nx = cx / cw, ny = cy / cw, nz = cz / cw;
    const float ipx = ((nx + 1.0f) * 0.5f - 1.0f * (1.0f - tf)) * (float)(fr->width - 1);
    const float ipy = ((ny + 1.0f) * 0.5f - 0.75f * (1.0f - tf)) * (float)(fr->height - 1);

    /* :190-194, get_rect :144-148 */
    const float radius = (gso_pow02(opacity) * 2.5f) * sqrtf(fmaxf(l1, l2));
    const float gxf = (float)gx, gyf = (float)gy;
    uint32_t x0 = (uint32_t)(int32_t)clampf((ipx - radius) / 16.0f, 0.0f, gxf);
    uint32_t y0 = (uint32_t)(int32_t)clampf((ipy - radius) / 16.0f, 0.0f, gyf);
    uint32_t x1 = (uint32_t)(int32_t)clampf(ceilf((ipx + radius) / 16.0f), 0.0f, gxf);
    uint32_t y1 = (uint32_t)(int32_t)clampf(ceilf((ipy + radius) / 16.0f), 0.0f, gyf);
    /* multi-GPU shard: keep only the tiles of this context's stripe.  The last tile of the
     * unclamped rectangle is recorded first: every shard projects every splat, so each one can tell whether
     * its highest populated tile is also the whole frame's (quirk Q5/Q6 must hit that tile only). */
    o->last_tile_plus1 = (x1 > x0 && y1 > y0) ? (y1 - 1) * gx + (x1 - 1) + 1 : 0;
    o->count = 0;
    if (x0 < fr->stripe_x0) x0 = fr->stripe_x0;
    if (y0 < fr->stripe_y0) y0 = fr->stripe_y0;
    if (x1 > fr->stripe_x1) x1 = fr->stripe_x1;
    if (y1 > fr->stripe_y1) y1 = fr->stripe_y1;
    if (x1 <= x0 || y1 <= y0) return 0;
    const uint32_t count = (x1 - x0) * (y1 - y0);

    /* :198-206 */
    const float dx = px - fr->cam_pos[0], dy = py - fr->cam_pos[1], dz = pz - fr->cam_pos[2];
    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float rgb[3];
    for (int ch = 0; ch < 3; ++ch) {
        const float *c = s + 12 + ch; /* SH_COEFFICIENTS(i) = c[3*i] */
        float v = 0.5f;
        v = v + c[0] * SH_C0;
        v = v - (c[3] * SH_C1) * y;
        v = v + (c[6] * SH_C1) * z;
        v = v - (c[9] * SH_C1) * x;
        v = v + (c[12] * SH_C2_0) * xy;
        v = v - (c[15] * SH_C2_1) * yz;
        v = v + (c[18] * SH_C2_2) * ((2.0f * zz - xx) - yy);
        v = v - (c[21] * SH_C2_3) * xz;
        v = v + (c[24] * SH_C2_4) * (xx - yy);
        v = v - ((c[27] * SH_C3_0) * y) * (3.0f * xx - yy);
        v = v + ((c[30] * SH_C3_1) * x) * yz;
        v = v - ((c[33] * SH_C3_2) * y) * ((4.0f * zz - xx) - yy);
        v = v + ((c[36] * SH_C3_3) * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
        v = v - ((c[39] * SH_C3_4) * x) * ((4.0f * zz - xx) - yy);
        v = v + ((c[42] * SH_C3_5) * z) * (xx - yy);
        v = v - ((c[45] * SH_C3_6) * x) * (xx - 3.0f * yy);
        rgb[ch] = fmaxf(0.0f, v);
    }
    float *r = o->raster;
    r[0] = ipx; r[1] = ipy;             /* image_pos */
    r[2] = px;  r[3] = py;              /* pos_xy */
    r[4] = cc / det; r[5] = (-cb) / det; r[6] = ca / det; /* conic */
    r[7] = pz;                          /* pos_z */
    r[8] = rgb[0]; r[9] = rgb[1]; r[10] = rgb[2]; r[11] = opacity;

    /* :218 */
    o->depth16 = (uint32_t)(((nz * nz) * nz) * 65535.0f) & 0xFFFFu;
    o->x0 = x0; o->y0 = y0; o->x1 = x1; o->y1 = y1;
    o->count = count;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * stage 1: projection + key emission  (gsplat_projection.glsl)
 * culled: n*12 floats (untouched for culled splats — the caller zero-fills, like a fresh buffer)
 * counts: n uint32 (num_tiles_touched, 0 for culled)
 * keys/values: capacity entries
 * returns D (un-clamped)
 * ---------------------------------------------------------------------------------------------- */
uint64_t gso_project(const float *splats, uint32_t n, const gso_frame *fr, uint64_t capacity,
                     float *culled, uint32_t *counts, uint32_t *keys, uint32_t *values,
                     uint64_t *visible_out, uint32_t *frame_last_tile_plus1) {
    const uint32_t gx = (uint32_t)(fr->width + GSO_TILE - 1) / GSO_TILE;
    const uint32_t gy = (uint32_t)(fr->height + GSO_TILE - 1) / GSO_TILE;
    uint32_t *rect = (uint32_t *)malloc((size_t)n * 5 * sizeof(uint32_t));
    uint64_t *offs = (uint64_t *)malloc(((size_t)n + 1) * sizeof(uint64_t));
    uint64_t visible = 0;
    uint32_t last_plus1 = 0;
#pragma omp parallel for schedule(static) reduction(+ : visible) reduction(max : last_plus1)
    for (int64_t id = 0; id < (int64_t)n; ++id) {
        gso_proj o;
        const int alive = project_one(splats + (size_t)id * 60, fr, gx, gy, &o);
        if (o.last_tile_plus1 > last_plus1) last_plus1 = o.last_tile_plus1;
        if (alive) {
            memcpy(culled + (size_t)id * 12, o.raster, 48);
            counts[id] = o.count;
            rect[id * 5 + 0] = o.x0; rect[id * 5 + 1] = o.y0;
            rect[id * 5 + 2] = o.x1; rect[id * 5 + 3] = o.y1;
            rect[id * 5 + 4] = o.depth16;
            visible += 1;
        } else {
            counts[id] = 0;
        }
    }
    /* :196 deterministic member: exclusive prefix sum over ascending id */
    uint64_t run = 0;
    for (uint32_t id = 0; id < n; ++id) { offs[id] = run; run += counts[id]; }
    offs[n] = run;
    /* :218-226 y-outer / x-inner duplication */
#pragma omp parallel for schedule(static)
    for (int64_t id = 0; id < (int64_t)n; ++id) {
        if (!counts[id]) continue;
        uint64_t off = offs[id];
        const uint32_t *r = rect + id * 5;
        for (uint32_t y = r[1]; y < r[3]; ++y)
            for (uint32_t x = r[0]; x < r[2]; ++x) {
                if (off < capacity) { /* never write past the budget */
                    keys[off] = ((y * gx + x) << 16) | r[4];
                    values[off] = (uint32_t)id;
                }
                ++off;
            }
    }
    free(rect);
    free(offs);
    if (visible_out) *visible_out = visible;
    if (frame_last_tile_plus1) *frame_last_tile_plus1 = last_plus1;
    return run;
}

/* ------------------------------------------------------------------------------------------------
 * stage 2: sort contract of radix_sort_{upsweep,spine,downsweep}.glsl — a stable LSD radix sort of
 * (key,value) pairs on the full 32-bit key, four 8-bit passes (radix_sort_downsweep.glsl:178-213:
 * dst = global[digit] + partition[digit] + local rank, i.e. stable).  Result in keys/values.
 * ---------------------------------------------------------------------------------------------- */
void gso_sort_pairs(uint32_t *keys, uint32_t *values, uint64_t d) {
    uint32_t *k2 = (uint32_t *)malloc((size_t)(d ? d : 1) * 4), *v2 = (uint32_t *)malloc((size_t)(d ? d : 1) * 4);
    uint32_t *ki = keys, *vi = values, *ko = k2, *vo = v2;
    for (int pass = 0; pass < 4; ++pass) {
        uint64_t hist[257];
        memset(hist, 0, sizeof hist);
        const int sh = 8 * pass;
        for (uint64_t i = 0; i < d; ++i) hist[((ki[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; ++b) hist[b + 1] += hist[b];
        for (uint64_t i = 0; i < d; ++i) {
            uint64_t dst = hist[(ki[i] >> sh) & 255u]++;
            ko[dst] = ki[i];
            vo[dst] = vi[i];
        }
        uint32_t *t;
        t = ki; ki = ko; ko = t;
        t = vi; vi = vo; vo = t;
    }
    /* 4 passes: data is back in keys/values (gaussian_splatting_rasterizer.gd:144-148 ping-pong) */
    free(k2);
    free(v2);
}

/* ------------------------------------------------------------------------------------------------
 * stage 3: tile ranges  (gsplat_boundaries.glsl:23-50), bounds pre-cleared to 0
 * (gaussian_splatting_rasterizer.gd:128).  Reproduces SURVEY Q5/Q6.
 * ---------------------------------------------------------------------------------------------- */
void gso_boundaries(const uint32_t *keys, uint64_t d, uint32_t num_tiles, uint32_t *bounds /* T*2 */,
                    int sharded, uint32_t frame_last_tile_plus1) {
    memset(bounds, 0, (size_t)num_tiles * 8);
    const uint32_t last = num_tiles - 1;
    for (uint64_t i = 1; i < d; ++i) { /* :27 id >= size || id == 0 -> return */
        const uint32_t prev = keys[i - 1] >> 16, cur = keys[i] >> 16;
        if (prev != cur) {
            bounds[2 * prev + 1] = (uint32_t)i; /* .y */
            bounds[2 * cur + 0] = (uint32_t)i;  /* .x */
        }
        if (cur == last) bounds[2 * last + 1] = (uint32_t)(d - 1); /* :47-49 */
    }
    /* Sharded frame: the quirk belongs to the highest populated tile of the WHOLE frame; a shard whose own
     * highest tile is a different one closes that tile's range normally. */
    if (sharded && d > 0) {
        const uint32_t t = keys[d - 1] >> 16;
        if (t + 1 != frame_last_tile_plus1) bounds[2 * t + 1] = (uint32_t)d;
    }
}

/* ------------------------------------------------------------------------------------------------
 * stage 4: tile compositor  (gsplat_render.glsl:50-111)
 * image: width*height*4 floats, row-major, y down.  pick: 4 floats (splat_pos.xyz, num_tile_splats),
 * written only under the condition of :105 (caller pre-clears).
 * exp_scale perturbs every exp() by a constant factor — 1.0f for the contract; tests use 1±eps to
 * find knife-edge pixels for the fast-exp kernel variant.
 * ---------------------------------------------------------------------------------------------- */
void gso_render(const float *culled, const uint32_t *values, const uint32_t *bounds, const gso_frame *fr,
                uint32_t tile_x0, uint32_t tile_x1, uint32_t tile_y0, uint32_t tile_y1, float exp_scale,
                float *image, float *pick, gso_stats *stats) {
    const int W = fr->width, H = fr->height;
    const uint32_t gx = (uint32_t)(W + GSO_TILE - 1) / GSO_TILE;
    const float MIN_ALPHA = 1.0f / 255.0f; /* :7 */
    uint64_t composited = 0, evals = 0, wave_steps = 0, wave_full = 0;
    const int64_t ntx = (int64_t)tile_x1 - tile_x0, nty = (int64_t)tile_y1 - tile_y0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : composited, evals, wave_steps, wave_full)
    for (int64_t ti = 0; ti < ntx * nty; ++ti) {
        const uint32_t bx = tile_x0 + (uint32_t)(ti % ntx), by = tile_y0 + (uint32_t)(ti / ntx);
        const uint32_t tile_id = by * gx + bx;
        const uint32_t b0 = bounds[2 * tile_id], b1 = bounds[2 * tile_id + 1];
        int32_t num = (int32_t)(b1 - b0); /* :61 */
        if (num < 0) num = 0;
        const int iters = (num + GSO_BLOCK - 1) / GSO_BLOCK; /* :62 */
        float Cr[GSO_BLOCK], Cg[GSO_BLOCK], Cb[GSO_BLOCK], T[GSO_BLOCK];
        for (int p = 0; p < GSO_BLOCK; ++p) { Cr[p] = Cg[p] = Cb[p] = 0.0f; T[p] = 1.0f; }
        uint32_t shared_t = ~0u; /* :51 */
        float st[GSO_BLOCK][9];
        for (int i = 0; i < iters && shared_t > 255u; ++i) { /* :66 */
            const int off = GSO_BLOCK * i;
            const int chunk = (num - off) < GSO_BLOCK ? (num - off) : GSO_BLOCK; /* :68 */
            for (int j = 0; j < chunk; ++j) { /* :72-75 staging (entries past the range are never read) */
                const float *r = culled + (size_t)values[(size_t)b0 + off + j] * 12;
                st[j][0] = r[0]; st[j][1] = r[1];
                st[j][2] = (-0.5f * r[4]) * GSO_LOG2E; /* hx */
                st[j][3] = (-r[5]) * GSO_LOG2E;        /* hy */
                st[j][4] = (-0.5f * r[6]) * GSO_LOG2E; /* hz */
                st[j][5] = r[11];
                st[j][6] = r[8]; st[j][7] = r[9]; st[j][8] = r[10];
            }
            composited += (uint64_t)chunk;
            shared_t = 0; /* :76 */
            unsigned char live[4][GSO_BLOCK], full[4][GSO_BLOCK]; /* per wave64 (4 pixel rows) and staged splat */
            memset(live, 0, sizeof live);
            memset(full, 0, sizeof full);
            for (int p = 0; p < GSO_BLOCK; ++p) {
                const float pxf = (float)(bx * GSO_TILE + (uint32_t)(p % GSO_TILE)); /* :58 */
                const float pyf = (float)(by * GSO_TILE + (uint32_t)(p / GSO_TILE));
                float t = T[p], cr = Cr[p], cg = Cg[p], cb = Cb[p];
                int j = 0;
                for (; j < chunk && t > MIN_ALPHA; ++j) { /* :79 */
                    const float dx = st[j][0] - pxf, dy = st[j][1] - pyf; /* :82 */
                    float a1 = st[j][2] * dx;
                    a1 = fmaf(st[j][3], dy, a1);
                    const float a2 = st[j][4] * dy;
                    float y = a2 * dy;
                    y = fmaf(a1, dx, y); /* :84 power * log2(e) */
                    live[GSO_WAVE_OF(p)][j] = 1;
                    if (!(y >= GSO_EXP_CUTOFF)) continue; /* contract: exp underflows to exactly 0 below the cutoff */
                    full[GSO_WAVE_OF(p)][j] = 1;
                    const float alpha = st[j][5] * (gso_exp2(y) * exp_scale); /* :86 */
                    const float w = alpha * t;
                    cr = fmaf(st[j][6], w, cr); /* :89 */
                    cg = fmaf(st[j][7], w, cg);
                    cb = fmaf(st[j][8], w, cb);
                    t = t - w; /* :90 */
                }
                evals += (uint64_t)j;
                T[p] = t; Cr[p] = cr; Cg[p] = cg; Cb[p] = cb;
                shared_t += (uint32_t)(t * 255.0f); /* :97 */
            }
            for (int w = 0; w < 4; ++w)
                for (int j = 0; j < chunk; ++j) { wave_steps += live[w][j]; wave_full += full[w][j]; }
        }
        /* :100-101 */
        const float a = (float)num * 5e-4f;
        const float h0 = 0.0f * (1.0f - a) + 1.0f * a;
        const float h1 = 0.0f * (1.0f - a) + 0.2f * a;
        const float h2 = 1.0f * (1.0f - a) + 0.2f * a;
        int wrote_pick = 0;
        for (int p = 0; p < GSO_BLOCK; ++p) {
            const int ix = (int)(bx * GSO_TILE) + p % GSO_TILE, iy = (int)(by * GSO_TILE) + p / GSO_TILE;
            const float om = 1.0f - T[p];
            if (ix < W && iy < H) {
                float *o = image + ((size_t)iy * W + ix) * 4;
                o[0] = Cr[p] + (h0 * om) * fr->heatmap_factor;
                o[1] = Cg[p] + (h1 * om) * fr->heatmap_factor;
                o[2] = Cb[p] + (h2 * om) * fr->heatmap_factor;
                o[3] = 1.0f;
            }
            /* :105 subgroupElect(): first invocation of each subgroup.  The reference's sort only
             * works with 32-wide subgroups, so "elected" = local index % 32 == 0. */
            if ((p & 31) == 0 && tile_id == fr->target_tile && T[p] != 1.0f) wrote_pick = 1;
        }
        if (wrote_pick && pick) {
            const float *r = culled + (size_t)values[(size_t)b0 + (b1 - b0) / 10u] * 12; /* :107 */
            pick[0] = r[2]; pick[1] = r[3]; pick[2] = r[7]; pick[3] = (float)num;
        }
    }
    if (stats) { stats->composited = composited; stats->evals = evals; stats->wave_steps = wave_steps; stats->wave_full = wave_full; }
}

/* ------------------------------------------------------------------------------------------------
 * whole frame = rasterize() of gaussian_splatting_rasterizer.gd:122-160.
 * All output pointers are caller-allocated: culled n*12 f32 (pre-zeroed), counts n, keys/values
 * capacity each, bounds T*2, image W*H*4, pick 4 (pre-cleared).  keys_unsorted/values_unsorted may be
 * NULL; if given they receive the emission-order pairs.
 * ---------------------------------------------------------------------------------------------- */
int gso_frame_render(const float *splats, uint32_t n, const gso_frame *fr, uint64_t capacity, float *culled,
                     uint32_t *counts, uint32_t *keys, uint32_t *values, uint32_t *keys_unsorted,
                     uint32_t *values_unsorted, uint32_t *bounds, float *image, float *pick, gso_stats *stats) {
    const uint32_t gx = (uint32_t)(fr->width + GSO_TILE - 1) / GSO_TILE;
    const uint32_t gy = (uint32_t)(fr->height + GSO_TILE - 1) / GSO_TILE;
    uint64_t visible = 0;
    uint32_t last_plus1 = 0;
    const uint64_t d_all = gso_project(splats, n, fr, capacity, culled, counts, keys, values, &visible, &last_plus1);
    const uint64_t d = d_all < capacity ? d_all : capacity;
    if (keys_unsorted) memcpy(keys_unsorted, keys, (size_t)d * 4);
    if (values_unsorted) memcpy(values_unsorted, values, (size_t)d * 4);
    gso_sort_pairs(keys, values, d);
    const int sharded = fr->stripe_x0 > 0 || fr->stripe_y0 > 0 || fr->stripe_x1 < gx || fr->stripe_y1 < gy;
    gso_boundaries(keys, d, gx * gy, bounds, sharded, last_plus1);
    gso_stats local;
    memset(&local, 0, sizeof local);
    uint32_t sx0 = fr->stripe_x0, sx1 = fr->stripe_x1 < gx ? fr->stripe_x1 : gx;
    uint32_t sy0 = fr->stripe_y0, sy1 = fr->stripe_y1 < gy ? fr->stripe_y1 : gy;
    if (image) gso_render(culled, values, bounds, fr, sx0, sx1, sy0, sy1, 1.0f, image, pick, &local);
    if (stats) {
        stats->visible = visible;
        stats->emitted = d_all;
        stats->sorted = d;
        stats->composited = local.composited;
        stats->evals = local.evals;
        stats->wave_steps = local.wave_steps;
        stats->wave_full = local.wave_full;
        stats->overflow = d_all > capacity;
        uint32_t t = gx * gy, bits = 0;
        while ((1u << bits) < t) ++bits;
        stats->sig_bits = 16 + (int)bits;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * loader transform of ply_file.gd:41-69: one INRIA 62-float row -> one 60-float Splat record.
 * GDScript evaluates exp() and the sigmoid in binary64 (Variant float) and stores binary32; Basis /
 * Quaternion math is Godot 4.3 core (third-party, not in /root/reference; real_t = binary32):
 * Basis(Quaternion) = the rotation matrix with s = 2/|q|^2, so un-normalised quaternions are
 * implicitly normalised.  cov = (S*R^T)^T * (S*R^T) = R * S^2 * R^T.
 * ---------------------------------------------------------------------------------------------- */
void gso_ply_row_to_record(const float *p, float load_time, float *out) {
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    out[3] = load_time;
    const float sx = (float)exp((double)p[55]), sy = (float)exp((double)p[56]), sz = (float)exp((double)p[57]);
    const float qx = p[59], qy = p[60], qz = p[61], qw = p[58]; /* Quaternion(x=rot_1,y=rot_2,z=rot_3,w=rot_0) */
    const float d = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
    const float s = 2.0f / d;
    const float xs = qx * s, ys = qy * s, zs = qz * s;
    const float wx = qw * xs, wy = qw * ys, wz = qw * zs;
    const float xx = qx * xs, xy = qx * ys, xz = qx * zs;
    const float yy = qy * ys, yz = qy * zs, zz = qz * zs;
    /* rows of R */
    const float R[3][3] = {{1.0f - (yy + zz), xy - wz, xz + wy},
                           {xy + wz, 1.0f - (xx + zz), yz - wx},
                           {xz - wy, yz + wx, 1.0f - (xx + yy)}};
    /* M = S * R^T : M[i][j] = s_i * R[j][i];  cov = M^T * M : cov[i][j] = sum_k M[k][i]*M[k][j] */
    const float sc[3] = {sx, sy, sz};
    float M[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = sc[i] * R[j][i];
    float cov[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cov[i][j] = (M[0][i] * M[0][j] + M[1][i] * M[1][j]) + M[2][i] * M[2][j];
    /* ply_file.gd:54-59: x[0], y[0], z[0], y[1], z[1], z[2] of the Basis columns */
    out[4] = cov[0][0]; out[5] = cov[0][1]; out[6] = cov[0][2];
    out[7] = cov[1][1]; out[8] = cov[1][2]; out[9] = cov[2][2];
    out[10] = (float)(1.0 / (1.0 + exp(-(double)p[54]))); /* :62 */
    out[11] = 0.0f;
    for (int k = 0; k < 3; ++k) out[12 + k] = p[6 + k]; /* :65 f_dc */
    for (int k = 0; k < 45; k += 3) {                   /* :66-69 f_rest re-interleave */
        out[15 + k + 0] = p[9 + k / 3 + 0];
        out[15 + k + 1] = p[9 + k / 3 + 15];
        out[15 + k + 2] = p[9 + k / 3 + 30];
    }
}

void gso_ply_rows_to_records(const float *rows, uint32_t n, float load_time, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) gso_ply_row_to_record(rows + (size_t)i * 62, load_time, out + (size_t)i * 60);
}

/* ------------------------------------------------------------------------------------------------
 * update_camera_matrices() of gaussian_splatting_rasterizer.gd:175-195.
 * cam: camera-to-world transform after basis_override, 12 floats = basis columns X,Y,Z then origin O.
 * proj4: Godot Projection columns (16 floats).  out32 = view(16) | proj(16), column-major.
 * ---------------------------------------------------------------------------------------------- */
void gso_pack_camera(const float *cam, const float *proj4, float *out32) {
    const float *X = cam, *Y = cam + 3, *Z = cam + 6, *O = cam + 9;
    float *v = out32, *p = out32 + 16;
    v[0] = -X[0]; v[1] = Y[0];  v[2] = -Z[0]; v[3] = 0.0f;
    v[4] = -X[1]; v[5] = Y[1];  v[6] = -Z[1]; v[7] = 0.0f;
    v[8] = X[2];  v[9] = -Y[2]; v[10] = Z[2]; v[11] = 0.0f;
    v[12] = -((O[0] * X[0] + O[1] * X[1]) + O[2] * X[2]);
    v[13] = -((O[0] * -Y[0] + O[1] * -Y[1]) + O[2] * -Y[2]);
    v[14] = -((O[0] * Z[0] + O[1] * Z[1]) + O[2] * Z[2]);
    v[15] = 1.0f;
    for (int c = 0; c < 4; ++c) {
        p[c * 4 + 0] = proj4[c * 4 + 0];
        p[c * 4 + 1] = proj4[c * 4 + 1];
        p[c * 4 + 2] = proj4[c * 4 + 2];
    }
    p[3] = 0.0f; p[7] = 0.0f; p[11] = -1.0f; p[15] = 0.0f;
}

int gso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void gso_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* array forms of the contract math, for tests */
void gso_pow02_array(const float *x, float *out, uint64_t n) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = gso_pow02(x[i]);
}
/* the same over the floats whose bit patterns are first_bits, first_bits + 1, ... (the exhaustive GPU test) */
void gso_pow02_bits(uint32_t first_bits, float *out, uint64_t n) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const uint32_t b = first_bits + (uint32_t)i;
        float x;
        memcpy(&x, &b, 4);
        out[i] = gso_pow02(x);
    }
}
void gso_exp2_array(const float *x, float *out, uint64_t n) { for (uint64_t i = 0; i < n; ++i) out[i] = gso_exp2(x[i]); }
