// The per-splat arithmetic of gsplat_projection.glsl:150-206 in pieces — shared by the projection kernel (every splat:
// cull, footprint, tile rectangle, depth code) and by the compositor, which in a "lazy" frame recomputes the screen-space
// record of the splats it stages from the scene instead of reading a RasterizeData record back (raster.hip).  One
// expression per quantity, so who evaluated it cannot be seen in the output.  Arithmetic contract: IEEE
// binary32, no contraction (-ffp-contract=off), sums left to right, correctly rounded / and sqrt.
#pragma once
#include "gsplat_internal.h"

namespace gsplat {

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ __forceinline__ float ease_out_cubic(float x) {  // gsplat_projection.glsl:87-90
    const float a = 1.0f - x;
    return 1.0f - (a * a) * a;
}

// :152-166: model position, view position, clip position
struct ClipPos {
    float px, py, pz;      // position * model_scale
    float vx, vy, vz;      // view space
    float cx, cy, cz, cw;  // clip space
};
__device__ __forceinline__ ClipPos splat_clip(const FrameParams &fp, const float4 pt) {
    const float *V = fp.V, *P = fp.P;
    const float ms = fp.model_scale;
    ClipPos c;
    c.px = pt.x * ms; c.py = pt.y * ms; c.pz = pt.z * ms;
    c.vx = ((V[0] * c.px + V[4] * c.py) + V[8] * c.pz) + V[12];
    c.vy = ((V[1] * c.px + V[5] * c.py) + V[9] * c.pz) + V[13];
    c.vz = ((V[2] * c.px + V[6] * c.py) + V[10] * c.pz) + V[14];
    const float vw = ((V[3] * c.px + V[7] * c.py) + V[11] * c.pz) + V[15];
    c.cx = ((P[0] * c.vx + P[4] * c.vy) + P[8] * c.vz) + P[12] * vw;
    c.cy = ((P[1] * c.vx + P[5] * c.vy) + P[9] * c.vz) + P[13] * vw;
    c.cz = ((P[2] * c.vx + P[6] * c.vy) + P[10] * c.vz) + P[14] * vw;
    c.cw = ((P[3] * c.vx + P[7] * c.vy) + P[11] * c.vz) + P[15] * vw;
    return c;
}
__device__ __forceinline__ bool splat_outside_frustum(const ClipPos &c) {  // :160-166
    const float vb = c.cw * 1.2f;
    return (c.cx < -vb) || (c.cy < -vb) || (c.cz < 0.0f) || (c.cx > vb) || (c.cy > vb) || (c.cz > c.cw);
}

// :169-174 load animation + :124-142 project_covariance: (a, b, c) of the 2-D covariance with the 0.3 low-pass, its
// determinant, the animated opacity and the position animation factor tf
struct Footprint {
    float ca, cb, cc, det, opacity, tf;
};
__device__ __forceinline__ Footprint splat_footprint(const FrameParams &fp, const ClipPos &c, float load_time,
                                                     const float4 A, const float4 Bc) {
    const float *V = fp.V;
    const float ms = fp.model_scale;
    Footprint f;
    const float st = fp.time - load_time;
    f.tf = ease_out_cubic(clampf(st, 0.0f, 1.0f));
    const float tfl = ease_out_cubic(clampf(st - 0.35f, 0.0f, 1.0f));
    f.opacity = (Bc.z * tfl) * tfl;
    const float smod = ms * (2.0f * (1.0f - tfl) + 1.0f * tfl);
    const float C00 = (A.x * smod) * smod, C01 = (A.y * smod) * smod, C02 = (A.z * smod) * smod;
    const float C11 = (A.w * smod) * smod, C12 = (Bc.x * smod) * smod, C22 = (Bc.y * smod) * smod;
    // (fp.focal0 = (dims * 0.5) * tan_fov_inv, fp.lim = (1 / tan_fov_inv) * 1.3: per-frame constants, api.hip)
    const float zinv = 1.0f / c.vz;
    const float fx = fp.focal0_x * zinv;
    const float fy = fp.focal0_y * zinv;
    const float mx = clampf(c.vx * zinv, -fp.lim_x, fp.lim_x);
    const float my = clampf(c.vy * zinv, -fp.lim_y, fp.lim_y);
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
    f.ca = ((T00 * b0[0] + T01 * b0[1]) + T02 * b0[2]) + 0.3f;
    f.cb = (T10 * b0[0] + T11 * b0[1]) + T12 * b0[2];
    f.cc = ((T10 * b1[0] + T11 * b1[1]) + T12 * b1[2]) + 0.3f;
    f.det = f.ca * f.cc - f.cb * f.cb;  // :177
    return f;
}

// :184-185 image position (with the load animation's slide-in)
__device__ __forceinline__ void splat_image_pos(const FrameParams &fp, const ClipPos &c, float tf, float &ipx,
                                                float &ipy) {
    const float nx = c.cx / c.cw, ny = c.cy / c.cw;
    ipx = ((nx + 1.0f) * 0.5f - 1.0f * (1.0f - tf)) * fp.Wm1;
    ipy = ((ny + 1.0f) * 0.5f - 0.75f * (1.0f - tf)) * fp.Hm1;
}

// :202-206 the geometry half of RasterizeData: {image_pos, pos.xy} {conic, pos.z}
__device__ __forceinline__ void splat_raster_geometry(const ClipPos &c, const Footprint &f, float ipx, float ipy,
                                                      float4 &r0, float4 &r1) {
    r0 = make_float4(ipx, ipy, c.px, c.py);
    r1 = make_float4(f.cc / f.det, (-f.cb) / f.det, f.ca / f.det, c.pz);
}

// The form in which the compositor stages a splat's geometry (raster.hip: s_rec; blend_list reads it): the centre, the conic
// pre-multiplied into (hx, hy, hz) = (-0.5 cx, -cy, -0.5 cz) * log2(e), the animated opacity — 24 bytes in two float4.
// One expression for whoever produces it: the compositor itself from a RasterizeData record (eager frames) or from the
// scene (lazy frames), or the projection kernel of a "geometry-eager" frame, which writes these 32 bytes per visible
// splat so that the issue-bound compositor only gathers them (GEO frames, api.hip).
constexpr float STAGE_LOG2E = 0x1.715476p+0f;
__device__ __forceinline__ void staged_geometry(const float4 r0, const float4 r1, float opacity, float4 &g0, float4 &g1) {
    g0 = make_float4(r0.x, r0.y, (-0.5f * r1.x) * STAGE_LOG2E, (-r1.y) * STAGE_LOG2E);
    g1 = make_float4((-0.5f * r1.z) * STAGE_LOG2E, opacity, 0.0f, 0.0f);
}

}  // namespace gsplat
