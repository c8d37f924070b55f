// pmc_calibrate — known-byte-count micro-kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (and the raw
// TCC_EA0_RDREQ* counters they derive from) on the access patterns of this library: wide streaming reads, gathers of
// 48-byte RasterizeData records and of 192-byte SH blocks at random indices, streaming writes and 12-byte writes at a
// 48-byte stride.  Run it under `rocprofv3 --pmc ... --kernel-trace` (tools/pmc_calibrate.sh); the footprint (3 GiB)
// is far beyond the 256 MiB Infinity Cache.  Prints the bytes each kernel asks for, per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void cal_stream_read(const float4 *src, size_t n, float4 *sink) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}
__global__ void cal_stream_write(float4 *dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}
// one lane per gathered record of REC float4s (48 B: REC = 3, 192 B: REC = 12) at a pseudo-random record index
template <int REC>
__global__ void cal_gather(const float4 *src, uint32_t records, uint32_t count, float4 *sink) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t idx = (uint32_t)(((uint64_t)i * 2654435761ull + 12345ull) % records);
    const float4 *r = src + (size_t)idx * REC;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < REC; ++k) { const float4 v = r[k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x == 12345.678f) sink[0] = acc;
}
// 12 bytes written into every 48-byte record (the colour slots of RasterizeData), consecutive records
__global__ void cal_strided_write12(float *dst, uint32_t records) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= records) return;
    float *c = dst + (size_t)i * 12 + 8;
    c[0] = 1.0f; c[1] = 2.0f; c[2] = 3.0f;
}

int main() {
    const size_t bytes = 3ull << 30;  // 3 GiB
    float4 *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    const size_t n4 = bytes / 16;
    const uint32_t gathers = 4u << 20;  // 4 Mi records per gather launch
    const uint32_t rec48 = (uint32_t)(bytes / 48), rec192 = (uint32_t)(bytes / 192);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(cal_stream_read, dim3(4096), dim3(256), 0, 0, buf, n4, sink);
        hipLaunchKernelGGL(cal_gather<3>, dim3(gathers / 256), dim3(256), 0, 0, buf, rec48, gathers, sink);
        hipLaunchKernelGGL(cal_gather<12>, dim3(gathers / 256), dim3(256), 0, 0, buf, rec192, gathers, sink);
        hipLaunchKernelGGL(cal_stream_write, dim3(4096), dim3(256), 0, 0, buf, n4);
        hipLaunchKernelGGL(cal_strided_write12, dim3((rec48 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<float *>(buf), rec48);
    }
    CHECK(hipDeviceSynchronize());
    printf("{\"cal_stream_read\": {\"read\": %zu}, \"cal_gather<3>\": {\"read\": %zu, \"records\": %u, \"record_bytes\": 48}, "
           "\"cal_gather<12>\": {\"read\": %zu, \"records\": %u, \"record_bytes\": 192}, \"cal_stream_write\": {\"write\": %zu}, "
           "\"cal_strided_write12\": {\"write\": %zu, \"records\": %u}}\n",
           bytes, (size_t)gathers * 48, gathers, (size_t)gathers * 192, gathers, bytes, (size_t)rec48 * 12, rec48);
    return 0;
}
