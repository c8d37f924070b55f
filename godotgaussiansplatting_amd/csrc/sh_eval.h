// SH colour of a splat as seen from the camera — get_color, gsplat_projection.glsl:94-121 (+ the view direction of
// :198-199).  Shared by the projection kernel (band-0 scenes), the colour pass (bands 1..3: the splats the compositor
// is expected to stage), the compositor's fallback for a splat the colour pass did not predict, and the parity tap that
// fills the colour of every visible splat on demand — one expression, so who evaluates a colour cannot be seen.  Arithmetic contract: IEEE binary32, no contraction, sums left to right per channel.
#pragma once
#include "gsplat_internal.h"

namespace gsplat {

// gsplat_projection.glsl:6-21
constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f;
constexpr float SH_C2_1 = 1.0925484305920792f;
constexpr float SH_C2_2 = 0.31539156525252005f;
constexpr float SH_C2_3 = 1.0925484305920792f;
constexpr float SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = 0.5900435899266435f;
constexpr float SH_C3_1 = 2.890611442640554f;
constexpr float SH_C3_2 = 0.4570457994644658f;
constexpr float SH_C3_3 = 0.3731763325901154f;
constexpr float SH_C3_4 = 0.4570457994644658f;
constexpr float SH_C3_5 = 1.445305721320277f;
constexpr float SH_C3_6 = 0.5900435899266435f;

// get_color for one channel.  c[i] = SH coefficient i of this channel; bands above DEG are not loaded: their
// coefficients are zero and each dropped term is an exact +-0.
template <int DEG>
__device__ __forceinline__ float sh_channel(const float *c, float x, float y, float z, float xx, float yy, float zz,
                                            float xy, float yz, float xz) {
    float v = 0.5f;
    v = v + c[0] * SH_C0;
    if (DEG >= 1) {
        v = v - (c[1] * SH_C1) * y;
        v = v + (c[2] * SH_C1) * z;
        v = v - (c[3] * SH_C1) * x;
    }
    if (DEG >= 2) {
        v = v + (c[4] * SH_C2_0) * xy;
        v = v - (c[5] * SH_C2_1) * yz;
        v = v + (c[6] * SH_C2_2) * ((2.0f * zz - xx) - yy);
        v = v - (c[7] * SH_C2_3) * xz;
        v = v + (c[8] * SH_C2_4) * (xx - yy);
    }
    if (DEG >= 3) {
        v = v - ((c[9] * SH_C3_0) * y) * (3.0f * xx - yy);
        v = v + ((c[10] * SH_C3_1) * x) * yz;
        v = v - ((c[11] * SH_C3_2) * y) * ((4.0f * zz - xx) - yy);
        v = v + ((c[12] * SH_C3_3) * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
        v = v - ((c[13] * SH_C3_4) * x) * ((4.0f * zz - xx) - yy);
        v = v + ((c[14] * SH_C3_5) * z) * (xx - yy);
        v = v - ((c[15] * SH_C3_6) * x) * (xx - 3.0f * yy);
    }
    return fmaxf(0.0f, v);
}

// Colour of a splat seen along the normalised direction (x, y, z) from its channel-major block of coefficients
// (SceneSoA::sh_block: channel ch = float4 4*ch .. 4*ch+3).  Two forms of the same arithmetic: channel after channel
// (16 coefficient registers: the compositor's cold fallback path must fit its 64-VGPR budget) and all loads first (the
// colour pass: every gather of the lane in flight at once).
template <int DEG>
__device__ __forceinline__ float sh_channel_of(const float4 *__restrict__ block, int ch, float x, float y, float z) {
    constexpr int NG = ((DEG + 1) * (DEG + 1) + 3) / 4;
    float c[16];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float4 v = block[4 * ch + g];
        c[4 * g] = v.x; c[4 * g + 1] = v.y; c[4 * g + 2] = v.z; c[4 * g + 3] = v.w;
    }
    return sh_channel<DEG>(c, x, y, z, x * x, y * y, z * z, x * y, y * z, x * z);
}

template <int DEG>
__device__ __forceinline__ void sh_rgb(const float4 *__restrict__ block, float x, float y, float z, float rgb[3]) {
    rgb[0] = sh_channel_of<DEG>(block, 0, x, y, z);
    rgb[1] = sh_channel_of<DEG>(block, 1, x, y, z);
    rgb[2] = sh_channel_of<DEG>(block, 2, x, y, z);
}

template <int DEG>
__device__ __forceinline__ void sh_rgb_wide(const float4 *__restrict__ block, float x, float y, float z, float rgb[3]) {
    constexpr int NG = ((DEG + 1) * (DEG + 1) + 3) / 4;
    float4 v[3][NG];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int g = 0; g < NG; ++g) v[ch][g] = block[4 * ch + g];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float c[16];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            c[4 * g] = v[ch][g].x; c[4 * g + 1] = v[ch][g].y; c[4 * g + 2] = v[ch][g].z; c[4 * g + 3] = v[ch][g].w;
        }
        rgb[ch] = sh_channel<DEG>(c, x, y, z, x * x, y * y, z * z, x * y, y * z, x * z);
    }
}

// view direction of get_color (gsplat_projection.glsl:198-199)
__device__ __forceinline__ void sh_direction(float px, float py, float pz, const float *cam, float &x, float &y,
                                             float &z) {
    const float dx = px - cam[0], dy = py - cam[1], dz = pz - cam[2];
    const float len = sqrtf((dx * dx + dy * dy) + dz * dz);
    x = dx / len; y = dy / len; z = dz / len;
}

}  // namespace gsplat
