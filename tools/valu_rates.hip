// valu_rates — issue cost of the instructions the compositor's blend loop and the projection are made of, on gfx950.
// For each instruction: 64 copies per loop trip on 8 independent accumulators (throughput form) or on one (dependent
// chain); 8192 workgroups of 4 waves, 8 waves per SIMD resident.  Prints cycles per wave-instruction per SIMD from the
// wall clock of the whole launch at 2.4 GHz (the same method as tools/step_rates.hip), and LDS read rates per CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_rates tools/valu_rates.hip ; run: tools/valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int TRIPS = 8192;  // loop trips (long enough that dispatching 8192 workgroups does not matter)
constexpr int REP = 8;       // copies of the 8-accumulator group per trip: 64 instructions per trip

struct Out { unsigned long long cycles; float sink; };

#define RATE_KERNEL(NAME, TYPE, INIT, ASM_INDEP, ...)                                                              \
    __global__ void NAME##_indep(Out *out, float seed) {                                                             \
        TYPE a0 = INIT(seed, 0), a1 = INIT(seed, 1), a2 = INIT(seed, 2), a3 = INIT(seed, 3), a4 = INIT(seed, 4),     \
             a5 = INIT(seed, 5), a6 = INIT(seed, 6), a7 = INIT(seed, 7);                                             \
        TYPE x = INIT(seed, 9), y = INIT(seed, 10);                                                                  \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int t = 0; t < TRIPS; ++t) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) {                                                        \
                asm volatile(ASM_INDEP : "+v"(a0) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a1) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a2) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a3) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a4) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a5) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a6) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
                asm volatile(ASM_INDEP : "+v"(a7) : "v"(x), "v"(y) : __VA_ARGS__);                                          \
            }                                                                                                        \
        }                                                                                                            \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        TYPE s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                              \
        if ((threadIdx.x & 63) == 0) {                                                                               \
            Out o; o.cycles = t1 - t0; o.sink = *reinterpret_cast<float *>(&s);                                      \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o;                                                   \
        }                                                                                                            \
    }                                                                                                                \
    __global__ void NAME##_dep(Out *out, float seed) {                                                               \
        TYPE a0 = INIT(seed, 0);                                                                                     \
        TYPE x = INIT(seed, 9), y = INIT(seed, 10);                                                                  \
        __syncthreads();                                                                                             \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                  \
        for (int t = 0; t < TRIPS; ++t) {                                                                            \
            _Pragma("unroll") for (int r = 0; r < REP * 8; ++r) asm volatile(ASM_INDEP : "+v"(a0) : "v"(x), "v"(y) : __VA_ARGS__); \
        }                                                                                                            \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                  \
        if ((threadIdx.x & 63) == 0) {                                                                               \
            Out o; o.cycles = t1 - t0; o.sink = *reinterpret_cast<float *>(&a0);                                     \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o;                                                   \
        }                                                                                                            \
    }

#define INIT_F(seed, k) ((seed) * (1.0f + 0.001f * (k)))
#define INIT_D(seed, k) ((double)(seed) * (1.0 + 0.001 * (k)))
#define INIT_F2(seed, k) (f2{(seed) * (1.0f + 0.001f * (k)), (seed) * (1.0f + 0.002f * (k))})
#define INIT_U(seed, k) ((unsigned)((seed) * 1000.0f) + (k))

RATE_KERNEL(fma_f32, float, INIT_F, "v_fma_f32 %0, %1, %2, %0", "memory")
RATE_KERNEL(fmac_f32, float, INIT_F, "v_fmac_f32_e32 %0, %1, %2", "memory")
RATE_KERNEL(mul_f32, float, INIT_F, "v_mul_f32_e32 %0, %1, %0", "memory")
RATE_KERNEL(add_f32_lit, float, INIT_F, "v_add_f32_e32 %0, 0x4b400000, %0", "memory")
RATE_KERNEL(fmaak_f32, float, INIT_F, "v_fmaak_f32 %0, %0, %1, 0x3d635ba9", "memory")
RATE_KERNEL(min_f32_lit, float, INIT_F, "v_min_f32 %0, 0x42fc0000, %0", "memory")
RATE_KERNEL(pk_fma_f32, f2, INIT_F2, "v_pk_fma_f32 %0, %1, %2, %0", "memory")
RATE_KERNEL(pk_fma_f32_opsel, f2, INIT_F2, "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]", "memory")
RATE_KERNEL(pk_mul_f32, f2, INIT_F2, "v_pk_mul_f32 %0, %1, %0", "memory")
RATE_KERNEL(cmp_vcc, float, INIT_F, "v_cmp_le_f32_e32 vcc, %1, %0", "vcc")
RATE_KERNEL(cmp_sgpr, float, INIT_F, "v_cmp_lt_f32_e64 s[20:21], %1, %0", "s20", "s21")
RATE_KERNEL(cndmask, float, INIT_F, "v_cndmask_b32_e32 %0, %1, %0, vcc", "memory")
RATE_KERNEL(lshl_add_u32, unsigned, INIT_U, "v_lshl_add_u32 %0, %1, 23, %0", "memory")
RATE_KERNEL(add_u32, unsigned, INIT_U, "v_add_u32_e32 %0, %1, %0", "memory")
RATE_KERNEL(mov_b32, float, INIT_F, "v_mov_b32_e32 %0, %1", "memory")
RATE_KERNEL(exp_f32, float, INIT_F, "v_exp_f32_e32 %0, %0", "memory")
RATE_KERNEL(log_f32, float, INIT_F, "v_log_f32_e32 %0, %0", "memory")
RATE_KERNEL(rcp_f32, float, INIT_F, "v_rcp_f32_e32 %0, %0", "memory")
RATE_KERNEL(sqrt_f32, float, INIT_F, "v_sqrt_f32_e32 %0, %0", "memory")
RATE_KERNEL(ldexp_f32, float, INIT_F, "v_ldexp_f32 %0, %0, %1", "memory")
RATE_KERNEL(rndne_f32, float, INIT_F, "v_rndne_f32_e32 %0, %0", "memory")
RATE_KERNEL(cvt_i32_f32, float, INIT_F, "v_cvt_i32_f32_e32 %0, %0", "memory")
RATE_KERNEL(mul_lo_u32, unsigned, INIT_U, "v_mul_lo_u32 %0, %1, %0", "memory")
RATE_KERNEL(fma_f64, double, INIT_D, "v_fma_f64 %0, %1, %2, %0", "memory")
RATE_KERNEL(mul_f64, double, INIT_D, "v_mul_f64 %0, %1, %0", "memory")
RATE_KERNEL(add_f64, double, INIT_D, "v_add_f64 %0, %1, %0", "memory")
RATE_KERNEL(div_scale_f32, float, INIT_F, "v_div_scale_f32 %0, vcc, %1, %2, %0", "vcc")
RATE_KERNEL(div_fmas_f32, float, INIT_F, "v_div_fmas_f32 %0, %1, %2, %0", "memory")
RATE_KERNEL(div_fixup_f32, float, INIT_F, "v_div_fixup_f32 %0, %1, %2, %0", "memory")
RATE_KERNEL(med3_f32, float, INIT_F, "v_med3_f32 %0, %1, %2, %0", "memory")
RATE_KERNEL(readlane_bcast, float, INIT_F, "v_readfirstlane_b32 s20, %0", "s20")

// LDS reads at one address for the whole wave (the blend loop's record reads) and at lane-consecutive addresses
template <int MODE>
__global__ void lds_read(Out *out, float seed) {
    __shared__ f4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = f4{seed, seed + i, 1.0f, 2.0f};
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t addr = (MODE == 2 ? (wave * 64 + lane) * 16 : MODE == 3 ? (wave * 64 + lane) * 4 : wave * 48) & 16383;
    f4 acc = f4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            f4 v0, v1, v2, v3, v4, v5, v6, v7;
            if (MODE == 0 || MODE == 2) {   // b128
                asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:48\n ds_read_b128 %2, %8 offset:96\n ds_read_b128 %3, %8 offset:144\n"
                             "ds_read_b128 %4, %8 offset:192\n ds_read_b128 %5, %8 offset:240\n ds_read_b128 %6, %8 offset:288\n ds_read_b128 %7, %8 offset:336\n s_waitcnt lgkmcnt(0)"
                             : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
                acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
            } else {                        // b32
                float w0, w1, w2, w3, w4, w5, w6, w7;
                asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:48\n ds_read_b32 %2, %8 offset:96\n ds_read_b32 %3, %8 offset:144\n"
                             "ds_read_b32 %4, %8 offset:192\n ds_read_b32 %5, %8 offset:240\n ds_read_b32 %6, %8 offset:288\n ds_read_b32 %7, %8 offset:336\n s_waitcnt lgkmcnt(0)"
                             : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3), "=v"(w4), "=v"(w5), "=v"(w6), "=v"(w7) : "v"(addr) : "memory");
                acc.x += w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { Out o; o.cycles = t1 - t0; o.sink = acc.x + acc.y + acc.z + acc.w; out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o; }
}
template <typename K>
static double run(K kernel, double instr_per_wave, Out *d_out) {
    const int wgs = 2048 * 4;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, d_out, 1.0f);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1e-3 * 2.4e9 * 1024.0 / ((double)wgs * 4 * instr_per_wave));
    }
    return best;
}

int main() {
    Out *d_out; CHECK(hipMalloc(&d_out, sizeof(Out) * 2048 * 4 * 4));
    const double K = (double)TRIPS * REP * 8;
    printf("| instruction | cycles per wave-instruction per SIMD, 8 waves per SIMD: independent | dependent chain |\n|---|---|---|\n");
#define ROW(NAME) printf("| %s | %.2f | %.2f |\n", #NAME, run(NAME##_indep, K, d_out), run(NAME##_dep, K, d_out)); fflush(stdout);
    ROW(fma_f32) ROW(fmac_f32) ROW(mul_f32) ROW(add_f32_lit) ROW(fmaak_f32) ROW(min_f32_lit) ROW(pk_fma_f32) ROW(pk_fma_f32_opsel)
    ROW(pk_mul_f32) ROW(cmp_vcc) ROW(cmp_sgpr) ROW(cndmask) ROW(lshl_add_u32) ROW(add_u32) ROW(mov_b32) ROW(exp_f32) ROW(log_f32)
    ROW(rcp_f32) ROW(sqrt_f32) ROW(ldexp_f32) ROW(rndne_f32) ROW(cvt_i32_f32) ROW(mul_lo_u32) ROW(fma_f64) ROW(mul_f64) ROW(add_f64)
    ROW(div_scale_f32) ROW(div_fmas_f32) ROW(div_fixup_f32) ROW(med3_f32) ROW(readlane_bcast)
    printf("\n| LDS read | cycles per wave-instruction per CU (32 waves per CU) |\n|---|---|\n");
    const double KL = (double)TRIPS * 64;
#define LROW(LABEL, KERN) printf("| %s | %.2f |\n", LABEL, run(KERN, KL, d_out) / 4); fflush(stdout);
    LROW("ds_read_b128, one address per wave (broadcast)", lds_read<0>)
    LROW("ds_read_b32, one address per wave (broadcast)", lds_read<1>)
    LROW("ds_read_b128, lane-consecutive 16 B", lds_read<2>)
    LROW("ds_read_b32, lane-consecutive 4 B", lds_read<3>)
    return 0;
}
