// Scene ingest for gfx950 — replaces the CPU swizzle + buffer_update loop of util/ply_file.gd:28-77.
//
// The reference converts every INRIA .ply row (62 floats) to a 240-byte AoS `Splat` record on worker
// threads and uploads the records.  Here the raw rows (or ready-made records) are copied to the GPU and
// a kernel writes the structure-of-arrays scene the kernels read (SceneSoA): band 0 of the SH
// coefficients as one float4 per splat, all 48 as a channel-major 192-byte block per splat.  The highest SH
// band with a non-zero coefficient is tracked so that bands which are all zero are never read.
// A workgroup stages its 128 rows through LDS: the source (device memory or the pinned staging ring of api.hip) is
// read as one contiguous, coalesced run whatever the row stride.
#include "gsplat_internal.h"
#include <cstring>

namespace gsplat {

namespace {

__device__ __forceinline__ uint32_t sh_degree_needed(const float *sh48) {
    uint32_t deg = 0;
#pragma unroll
    for (int i = 3; i < 48; ++i) {
        if (sh48[i] != 0.0f) {  // NaN counts as non-zero; -0.0 as zero (its terms are exact zeros)
            const uint32_t d = i < 12 ? 1u : (i < 27 ? 2u : 3u);
            deg = d > deg ? d : deg;
        }
    }
    return deg;
}

__device__ __forceinline__ void store_soa(const SceneSoA &scene, uint32_t id, const float *rec) {
    scene.pos_time[id] = make_float4(rec[0], rec[1], rec[2], rec[3]);
    scene.cov_a[id] = make_float4(rec[4], rec[5], rec[6], rec[7]);
    scene.cov_b[id] = make_float4(rec[8], rec[9], rec[10], rec[11]);
    // record float 12 + 3 i + ch = coefficient i of channel ch (struct Splat, gsplat_projection.glsl:39)
    scene.sh_dc[id] = make_float4(rec[12], rec[13], rec[14], 0.0f);
    // the compositor's slot holds a copy of the geometry behind the coefficients (gsplat_internal.h: SceneSoA).  A scene
    // that has only seen band-0 colours so far has no slots at all (api.hip: ensure_slots builds them from the planes
    // the moment a higher band arrives)
    if (scene.sh_block == nullptr) return;
    float4 *slot = scene.sh_block + (size_t)id * SH_BLOCK_F4;
    slot[SLOT_POS] = make_float4(rec[0], rec[1], rec[2], rec[3]);
    slot[SLOT_COV_A] = make_float4(rec[4], rec[5], rec[6], rec[7]);
    slot[SLOT_COV_B] = make_float4(rec[8], rec[9], rec[10], rec[11]);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            scene.sh_block[(size_t)id * SH_BLOCK_F4 + 4 * ch + g] =
                make_float4(rec[12 + 3 * (4 * g) + ch], rec[12 + 3 * (4 * g + 1) + ch], rec[12 + 3 * (4 * g + 2) + ch],
                            rec[12 + 3 * (4 * g + 3) + ch]);
}

constexpr int UPLOAD_BLOCK = 128;

// coalesced copy of this workgroup's `floats_per_item`-float items into LDS; returns the lane's item (or nullptr)
template <int FPI>
__device__ __forceinline__ const float *stage_items(const float *__restrict__ src, uint32_t count, float *lds) {
    const uint32_t first = blockIdx.x * UPLOAD_BLOCK;
    const uint32_t items = min((uint32_t)UPLOAD_BLOCK, count - first);
    const float *base = src + (size_t)first * FPI;
    for (uint32_t k = threadIdx.x; k < items * FPI; k += UPLOAD_BLOCK) lds[k] = base[k];
    __syncthreads();
    return threadIdx.x < items ? lds + threadIdx.x * FPI : nullptr;
}

// struct Splat records (gsplat_projection.glsl:33-40) -> SoA
__global__ __launch_bounds__(UPLOAD_BLOCK) void upload_records_kernel(SceneSoA scene, uint32_t first, uint32_t count,
                                                                      const float *__restrict__ records,
                                                                      uint32_t *__restrict__ sh_degree_max,
                                                                      const uint32_t *__restrict__ slot_of) {
    __shared__ float lds[UPLOAD_BLOCK * 60];
    const uint32_t i = blockIdx.x * UPLOAD_BLOCK + threadIdx.x;
    const float *rec = stage_items<60>(records, count, lds);
    uint32_t deg = 0;
    if (rec) {
        store_soa(scene, slot_of ? slot_of[first + i] : first + i, rec);  // re-laid-out scene: id -> slot
        deg = sh_degree_needed(rec + 12);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) deg = max(deg, (uint32_t)__shfl_xor((int)deg, d, 64));
    if ((threadIdx.x & 63) == 0 && deg) atomicMax(sh_degree_max, deg);
}

// ply_file.gd:41-69 on the GPU.  GDScript evaluates exp() and the sigmoid in binary64 and stores
// binary32; Basis/Quaternion math is Godot's binary32 real_t (Basis(Quaternion) divides by |q|^2).
__global__ __launch_bounds__(UPLOAD_BLOCK) void upload_ply_rows_kernel(SceneSoA scene, uint32_t first, uint32_t count,
                                                                       const float *__restrict__ rows, float load_time,
                                                                       uint32_t *__restrict__ sh_degree_max,
                                                                       const uint32_t *__restrict__ slot_of) {
    __shared__ float lds[UPLOAD_BLOCK * 62];
    const uint32_t i = blockIdx.x * UPLOAD_BLOCK + threadIdx.x;
    const float *p = stage_items<62>(rows, count, lds);
    uint32_t deg = 0;
    if (p) {
        float rec[60];
        rec[0] = p[0]; rec[1] = p[1]; rec[2] = p[2];
        rec[3] = load_time;
        const float sx = (float)exp((double)p[55]), sy = (float)exp((double)p[56]), sz = (float)exp((double)p[57]);
        const float qx = p[59], qy = p[60], qz = p[61], qw = p[58];
        const float d = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
        const float s = 2.0f / d;
        const float xs = qx * s, ys = qy * s, zs = qz * s;
        const float wx = qw * xs, wy = qw * ys, wz = qw * zs;
        const float xx = qx * xs, xy = qx * ys, xz = qx * zs;
        const float yy = qy * ys, yz = qy * zs, zz = qz * zs;
        const float R[3][3] = {{1.0f - (yy + zz), xy - wz, xz + wy},
                               {xy + wz, 1.0f - (xx + zz), yz - wx},
                               {xz - wy, yz + wx, 1.0f - (xx + yy)}};
        const float sc[3] = {sx, sy, sz};
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) M[a][b] = sc[a] * R[b][a];
        float cov[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) cov[a][b] = (M[0][a] * M[0][b] + M[1][a] * M[1][b]) + M[2][a] * M[2][b];
        rec[4] = cov[0][0]; rec[5] = cov[0][1]; rec[6] = cov[0][2];
        rec[7] = cov[1][1]; rec[8] = cov[1][2]; rec[9] = cov[2][2];
        rec[10] = (float)(1.0 / (1.0 + exp(-(double)p[54])));
        rec[11] = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) rec[12 + k] = p[6 + k];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            rec[15 + 3 * k + 0] = p[9 + k];
            rec[15 + 3 * k + 1] = p[9 + k + 15];
            rec[15 + 3 * k + 2] = p[9 + k + 30];
        }
        store_soa(scene, slot_of ? slot_of[first + i] : first + i, rec);
        deg = sh_degree_needed(rec + 12);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) deg = max(deg, (uint32_t)__shfl_xor((int)deg, d, 64));
    if ((threadIdx.x & 63) == 0 && deg) atomicMax(sh_degree_max, deg);
}

// SoA -> 60-float records (parity tap)
__global__ __launch_bounds__(256) void gather_records_kernel(SceneSoA scene, uint32_t n_total,
                                                             float *__restrict__ records,
                                                             const uint32_t *__restrict__ slot_of) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_total) return;
    const uint32_t slot = slot_of ? slot_of[id] : id;
    float4 *dst = reinterpret_cast<float4 *>(records + (size_t)id * 60);
    dst[0] = scene.pos_time[slot];
    dst[1] = scene.cov_a[slot];
    dst[2] = scene.cov_b[slot];
    float sh[48];
    const float4 dc = scene.sh_dc[slot];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (no slots: a band-0 scene — coefficient 0 from its plane, every higher coefficient is zero)
            const float4 v = scene.sh_block != nullptr ? scene.sh_block[(size_t)slot * SH_BLOCK_F4 + 4 * ch + g]
                             : make_float4(g == 0 ? (ch == 0 ? dc.x : (ch == 1 ? dc.y : dc.z)) : 0.0f, 0.0f, 0.0f, 0.0f);
            sh[3 * (4 * g) + ch] = v.x;
            sh[3 * (4 * g + 1) + ch] = v.y;
            sh[3 * (4 * g + 2) + ch] = v.z;
            sh[3 * (4 * g + 3) + ch] = v.w;
        }
#pragma unroll
    for (int p = 0; p < 12; ++p) dst[3 + p] = make_float4(sh[4 * p], sh[4 * p + 1], sh[4 * p + 2], sh[4 * p + 3]);
}

// The slots of a scene that held band-0 colours only until now: coefficient 0 from its plane, zeros above, the geometry
// copies from the planes — exactly what store_soa would have written for those splats.
__global__ __launch_bounds__(256) void build_slots_kernel(SceneSoA scene, uint32_t n) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    if (id >= n) return;
    float4 *slot = scene.sh_block + (size_t)id * SH_BLOCK_F4;
    const float4 dc = scene.sh_dc[id];
    const float c0[3] = {dc.x, dc.y, dc.z};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int g = 0; g < 4; ++g) slot[4 * ch + g] = make_float4(g == 0 ? c0[ch] : 0.0f, 0.0f, 0.0f, 0.0f);
    slot[SLOT_POS] = scene.pos_time[id];
    slot[SLOT_COV_A] = scene.cov_a[id];
    slot[SLOT_COV_B] = scene.cov_b[id];
    slot[15] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- gsplat_finalize_scene: 30-bit Morton codes of the positions, on the device --------------------------------------
// (round 2 copied all positions to the host and std::sort-ed N 64-bit words there: seconds at 30 M splats)
// floats as order-preserving unsigned keys, for atomicMin / atomicMax
__device__ __forceinline__ uint32_t float_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float key_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}

// box[0..2] = min key of x, y, z over the finite coordinates, box[3..5] = max key (initialised to ~0 / 0 by the host)
__global__ __launch_bounds__(256) void morton_bounds_kernel(const float4 *__restrict__ pos, uint32_t n,
                                                            uint32_t *__restrict__ box) {
    uint32_t lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 p = pos[i];
        const float c[3] = {p.x, p.y, p.z};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (isfinite(c[a])) {
                const uint32_t k = float_key(c[a]);
                lo[a] = min(lo[a], k);
                hi[a] = max(hi[a], k);
            }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], d, 64));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], d, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&box[a], lo[a]);
            atomicMax(&box[3 + a], hi[a]);
        }
    }
}

// code = 10 bits per axis of (p - lo) / (hi - lo), interleaved x | y << 1 | z << 2 per bit triple; value = splat id.
// Evaluated in binary64 like the host code it replaces: the same codes, hence the same layout.
__global__ __launch_bounds__(256) void morton_codes_kernel(const float4 *__restrict__ pos, uint32_t n,
                                                           const uint32_t *__restrict__ box,
                                                           uint32_t *__restrict__ codes, uint32_t *__restrict__ ids) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[i];
    const float c[3] = {p.x, p.y, p.z};
    uint32_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool any = box[a] <= box[3 + a];  // (no finite coordinate on this axis: lo stayed above hi)
        const double lo = any ? (double)key_float(box[a]) : 0.0, hi = any ? (double)key_float(box[3 + a]) : 0.0;
        double t = 0.0;
        if (isfinite(c[a]) && hi > lo) t = ((double)c[a] - lo) / (hi - lo);
        uint64_t q = (uint64_t)(t * 1023.0);
        if (q > 1023) q = 1023;
        uint64_t v = q;  // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FFull;
        v = (v | (v << 8)) & 0x0300F00Full;
        v = (v | (v << 4)) & 0x030C30C3ull;
        v = (v | (v << 2)) & 0x09249249ull;
        code |= (uint32_t)(v << a);
    }
    codes[i] = code;
    ids[i] = i;
}

__global__ __launch_bounds__(256) void invert_permutation_kernel(const uint32_t *__restrict__ id_of_slot, uint32_t n,
                                                                 uint32_t *__restrict__ slot_of_id) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot < n) slot_of_id[id_of_slot[slot]] = slot;
}

// Scene re-layout (gsplat_finalize_scene): dst[slot] = src[id_of[slot]] for an array of records of `rec` float4s
__global__ __launch_bounds__(256) void permute_float4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst,
                                                             const uint32_t *__restrict__ id_of, uint32_t n,
                                                             uint32_t rec) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per lane
    const uint64_t slot = i / rec;
    const uint32_t k = (uint32_t)(i - slot * rec);
    if (slot < n) dst[i] = src[(uint64_t)id_of[slot] * rec + k];
}

// parity taps of a re-laid-out scene: per-splat arrays back in splat-id order, sorted values back to splat ids
__global__ __launch_bounds__(256) void gather_u32_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                         const uint32_t *__restrict__ index, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

__global__ __launch_bounds__(256) void gather_raster_kernel(const float4 *__restrict__ culled, float4 *__restrict__ dst,
                                                            const uint32_t *__restrict__ slot_of, uint32_t n) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const float4 *s = culled + (size_t)slot_of[id] * 3;
    dst[(size_t)id * 3 + 0] = s[0];
    dst[(size_t)id * 3 + 1] = s[1];
    dst[(size_t)id * 3 + 2] = s[2];
}

}  // namespace

void launch_upload_records(const SceneSoA &scene, uint32_t n_total, uint32_t first, uint32_t count,
                           const float *d_records, uint32_t *sh_degree_max, const uint32_t *slot_of, hipStream_t s) {
    if (!count) return;
    (void)n_total;
    hipLaunchKernelGGL(upload_records_kernel, dim3((count + UPLOAD_BLOCK - 1) / UPLOAD_BLOCK), dim3(UPLOAD_BLOCK), 0, s,
                       scene, first, count, d_records, sh_degree_max, slot_of);
}

void launch_upload_ply_rows(const SceneSoA &scene, uint32_t n_total, uint32_t first, uint32_t count,
                            const float *d_rows, float load_time, uint32_t *sh_degree_max, const uint32_t *slot_of,
                            hipStream_t s) {
    if (!count) return;
    (void)n_total;
    hipLaunchKernelGGL(upload_ply_rows_kernel, dim3((count + UPLOAD_BLOCK - 1) / UPLOAD_BLOCK), dim3(UPLOAD_BLOCK), 0, s,
                       scene, first, count, d_rows, load_time, sh_degree_max, slot_of);
}

void launch_gather_records(const SceneSoA &scene, uint32_t n_total, float *d_records, const uint32_t *slot_of,
                           hipStream_t s) {
    if (!n_total) return;
    hipLaunchKernelGGL(gather_records_kernel, dim3((n_total + 255) / 256), dim3(256), 0, s, scene, n_total,
                       d_records, slot_of);
}

void launch_build_slots(const SceneSoA &scene, uint32_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(build_slots_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, scene, n);
}

void launch_morton_keys(const float4 *pos, uint32_t n, uint32_t *box6, uint32_t *codes, uint32_t *ids, hipStream_t s) {
    if (!n) return;
    (void)hipMemsetAsync(box6, 0xFF, 3 * sizeof(uint32_t), s);      // min keys start at the top,
    (void)hipMemsetAsync(box6 + 3, 0, 3 * sizeof(uint32_t), s);     // max keys at the bottom
    hipLaunchKernelGGL(morton_bounds_kernel, dim3(min((n + 255u) / 256u, 4096u)), dim3(256), 0, s, pos, n, box6);
    hipLaunchKernelGGL(morton_codes_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, pos, n, box6, codes, ids);
}

void launch_invert_permutation(const uint32_t *id_of_slot, uint32_t n, uint32_t *slot_of_id, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(invert_permutation_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, id_of_slot, n, slot_of_id);
}

void launch_permute_float4(const float4 *src, float4 *dst, const uint32_t *id_of, uint32_t n, uint32_t rec,
                           hipStream_t s) {
    if (!n) return;
    const uint64_t total = (uint64_t)n * rec;
    hipLaunchKernelGGL(permute_float4_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, src, dst, id_of, n,
                       rec);
}

void launch_gather_u32(const uint32_t *src, uint32_t *dst, const uint32_t *index, uint32_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(gather_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, index, n);
}

void launch_gather_raster(const float4 *culled, float4 *dst, const uint32_t *slot_of, uint32_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(gather_raster_kernel, dim3((n + 255) / 256), dim3(256), 0, s, culled, dst, slot_of, n);
}

}  // namespace gsplat
