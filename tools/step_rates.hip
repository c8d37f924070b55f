// step_rates — what one step of the compositor's blend loop costs a SIMD, instruction sequence by instruction sequence
// (gfx950).  Each kernel runs a fixed sequence (the ISA hipcc emits for raster.hip's loop, or a candidate replacement)
// TRIPS x 8 times per wave with W waves per SIMD on every CU and reports shader cycles per step per SIMD
// (median wave's s_memtime ticks / steps / W).  Records live in registers unless the variant says LDS.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/step_rates tools/step_rates.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TRIPS = 16384;
struct Out { unsigned long long t0, t1; float sink; };

// registers: v4..v7 = record {ipx, ipy, hx, hy}, v3 = hz, v30..v33 = {r, g, b, opacity}; v10, v11 = pixel; v12 = t;
// v20..v22 = colour sums; v14 = polynomial constant; s20 = cutoff, s21 = MIN_ALPHA; s[8:9] = alive mask
#define SETUP                                                                                                          \
    "v_mov_b32 v4, 0x41200000\n v_mov_b32 v5, 0x41300000\n v_mov_b32 v6, 0xbc23d70a\n v_mov_b32 v7, 0x3a83126f\n"       \
    "v_mov_b32 v3, 0xbc23d70a\n v_mov_b32 v30, 0x3f000000\n v_mov_b32 v31, 0x3e800000\n v_mov_b32 v32, 0x3e000000\n"    \
    "v_mov_b32 v33, 0x358637bd\n v_and_b32 v10, 7, v0\n v_cvt_f32_u32 v10, v10\n v_mov_b32 v11, 0x40400000\n v_mov_b32 v12, 1.0\n"          \
    "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v14, 0x3c1d955b\n v_mov_b32 v16, 0x3c1d955b\n v_mov_b32 v13, 0x40400000\n"                             \
    "s_mov_b32 s20, 0xc2000000\n s_mov_b32 s21, 0x3b808081\n s_mov_b64 s[8:9], exec\n s_mov_b64 vcc, exec\n"

#define EXPONENT                                                                                                       \
    "v_sub_f32_e32 v24, v4, v10\n v_sub_f32_e32 v25, v5, v11\n v_mul_f32_e32 v26, v24, v6\n v_mul_f32_e32 v23, v25, v3\n" \
    "v_fmac_f32_e32 v26, v7, v25\n v_mul_f32_e32 v23, v25, v23\n v_fmac_f32_e32 v23, v26, v24\n"

#define EXP2_POLY                                                                                                      \
    "v_min_f32 v0, 0x42fc0000, v23\n v_add_f32_e32 v1, 0x4b400000, v0\n v_add_f32_e32 v8, 0xcb400000, v1\n"            \
    "v_sub_f32_e32 v0, v0, v8\n v_fmamk_f32 v8, v0, 0x3aaddd0c, v14\n v_fmaak_f32 v8, v8, v0, 0x3d635ba9\n"            \
    "v_fmaak_f32 v8, v8, v0, 0x3e75fcde\n v_fmaak_f32 v8, v8, v0, 0x3f317215\n v_fma_f32 v0, v8, v0, 1.0\n"            \
    "v_lshl_add_u32 v0, v1, 23, v0\n"

#define ACCUM                                                                                                          \
    "v_mul_f32_e32 v0, v12, v0\n v_pk_fma_f32 v[20:21], v[30:31], v[0:1], v[20:21] op_sel_hi:[1,0,1]\n"                \
    "v_fmac_f32_e32 v22, v32, v0\n v_sub_f32_e32 v12, v12, v0\n"

// A: the loop as compiled today (seen step)
#define STEP_A                                                                                                         \
    EXPONENT "v_cmp_le_f32_e32 vcc, s20, v23\n s_and_b64 s[10:11], vcc, s[8:9]\n s_cmp_eq_u64 s[10:11], 0\n"          \
    "s_cbranch_scc1 1f\n s_and_saveexec_b64 s[10:11], s[8:9]\n s_cbranch_execz 2f\n v_cmp_le_f32_e32 vcc, s20, v23\n"  \
    EXP2_POLY "v_mul_f32_e32 v0, v33, v0\n v_cndmask_b32_e32 v0, 0, v0, vcc\n" ACCUM                                   \
    "2:\n s_or_b64 exec, exec, s[10:11]\n v_cmp_lt_f32_e64 s[8:9], s21, v12\n 1:\n"                                    \
    "s_cmp_eq_u64 s[8:9], 0\n s_cbranch_scc1 9f\n"
// B: lanes below the cutoff masked off through exec instead of v_cndmask; no second compare
#define STEP_B                                                                                                         \
    EXPONENT "v_cmp_le_f32_e32 vcc, s20, v23\n s_and_b64 s[10:11], vcc, s[8:9]\n s_cbranch_scc0 1f\n"                 \
    "s_and_saveexec_b64 s[12:13], s[10:11]\n"                                                                          \
    EXP2_POLY "v_mul_f32_e32 v0, v33, v0\n" ACCUM                                                                      \
    "s_mov_b64 exec, s[12:13]\n v_cmp_lt_f32_e64 s[8:9], s21, v12\n 1:\n"                                              \
    "s_cmp_eq_u64 s[8:9], 0\n s_cbranch_scc1 9f\n"
// C: exec IS the alive mask for the whole loop: v_cmp under it gives above & alive, v_cmpx retires pixels
#define STEP_C                                                                                                         \
    EXPONENT "v_cmp_le_f32_e32 vcc, s20, v23\n s_cbranch_vccz 1f\n s_and_saveexec_b64 s[12:13], vcc\n"                \
    EXP2_POLY "v_mul_f32_e32 v0, v33, v0\n" ACCUM                                                                      \
    "s_mov_b64 exec, s[12:13]\n v_cmpx_lt_f32_e32 s21, v12\n s_cbranch_execz 9f\n 1:\n"
// D: C with three scalar FMAs for the colour instead of packed + scalar
#define ACCUM3                                                                                                         \
    "v_mul_f32_e32 v0, v12, v0\n v_fmac_f32_e32 v20, v30, v0\n v_fmac_f32_e32 v21, v31, v0\n v_fmac_f32_e32 v22, v32, v0\n v_sub_f32_e32 v12, v12, v0\n"
#define STEP_D                                                                                                         \
    EXPONENT "v_cmp_le_f32_e32 vcc, s20, v23\n s_cbranch_vccz 1f\n s_and_saveexec_b64 s[12:13], vcc\n"                \
    EXP2_POLY "v_mul_f32_e32 v0, v33, v0\n" ACCUM3                                                                     \
    "s_mov_b64 exec, s[12:13]\n v_cmpx_lt_f32_e32 s21, v12\n s_cbranch_execz 9f\n 1:\n"
// E: a step that ends at the cutoff test (exponent + compare + branch taken), today's form and under exec = alive
#define STEP_E_TODAY                                                                                                   \
    EXPONENT "v_cmp_gt_f32_e32 vcc, s20, v23\n s_and_b64 s[10:11], vcc, s[8:9]\n s_cmp_eq_u64 s[10:11], 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n" \
    "s_cmp_eq_u64 s[8:9], 0\n s_cbranch_scc1 9f\n"
#define STEP_E_EXEC                                                                                                    \
    EXPONENT "v_cmp_gt_f32_e32 vcc, s20, v23\n s_cbranch_vccz 1f\n s_nop 0\n 1:\n"
// F: the exponent alone; G: the exp2 polynomial alone; H: accumulate alone
#define STEP_F EXPONENT
#define STEP_G EXP2_POLY
#define STEP_H ACCUM
// I: v_cndmask forms
#define STEP_I_VCC "v_cndmask_b32_e32 v0, 0, v23, vcc\n v_cndmask_b32_e32 v1, 0, v24, vcc\n v_cndmask_b32_e32 v8, 0, v25, vcc\n v_cndmask_b32_e32 v26, 0, v23, vcc\n"
#define STEP_I_SGPR "v_cndmask_b32_e64 v0, 0, v23, s[8:9]\n v_cndmask_b32_e64 v1, 0, v24, s[8:9]\n v_cndmask_b32_e64 v8, 0, v25, s[8:9]\n v_cndmask_b32_e64 v26, 0, v23, s[8:9]\n"
#define STEP_I_VV "v_cndmask_b32_e32 v0, v24, v23, vcc\n v_cndmask_b32_e32 v1, v25, v24, vcc\n v_cndmask_b32_e32 v8, v23, v25, vcc\n v_cndmask_b32_e32 v26, v24, v23, vcc\n"
// J: VALU with a scalar operand vs all-vector
#define STEP_J_S "v_mul_f32_e32 v0, s21, v23\n v_mul_f32_e32 v1, s21, v24\n v_mul_f32_e32 v8, s21, v25\n v_mul_f32_e32 v26, s21, v23\n"
#define STEP_J_V "v_mul_f32_e32 v0, v11, v23\n v_mul_f32_e32 v1, v11, v24\n v_mul_f32_e32 v8, v11, v25\n v_mul_f32_e32 v26, v11, v23\n"
#define STEP_J_CMP_S "v_cmp_le_f32_e32 vcc, s20, v23\n v_cmp_le_f32_e32 vcc, s20, v24\n v_cmp_le_f32_e32 vcc, s20, v25\n v_cmp_le_f32_e32 vcc, s20, v11\n"
#define STEP_J_CMP_V "v_cmp_le_f32_e32 vcc, v10, v23\n v_cmp_le_f32_e32 vcc, v10, v24\n v_cmp_le_f32_e32 vcc, v10, v25\n v_cmp_le_f32_e32 vcc, v10, v11\n"
#define STEP_J_CMPX "v_cmpx_lt_f32_e32 s21, v12\n v_cmpx_lt_f32_e32 s21, v12\n v_cmpx_lt_f32_e32 s21, v12\n v_cmpx_lt_f32_e32 s21, v12\n"
// K: max instead of cndmask for "0 below the cutoff" is not the same arithmetic; v_mul by a 0/1 float from v_cndmask is no cheaper.
// L: record reads from LDS as the loop does them, step C around them (one address per wave)
#define STEP_L                                                                                                         \
    "ds_read_b128 v[4:7], v40\n ds_read_b32 v3, v40 offset:16\n s_waitcnt lgkmcnt(1)\n"                                \
    "v_sub_f32_e32 v24, v4, v10\n v_sub_f32_e32 v25, v5, v11\n v_mul_f32_e32 v26, v24, v6\n s_waitcnt lgkmcnt(0)\n v_mul_f32_e32 v23, v25, v3\n" \
    "v_fmac_f32_e32 v26, v7, v25\n v_mul_f32_e32 v23, v25, v23\n v_fmac_f32_e32 v23, v26, v24\n"                       \
    "v_cmp_le_f32_e32 vcc, s20, v23\n s_cbranch_vccz 1f\n s_and_saveexec_b64 s[12:13], vcc\n ds_read_b128 v[30:33], v40 offset:32\n" \
    EXP2_POLY "s_waitcnt lgkmcnt(0)\n v_mul_f32_e32 v0, v33, v0\n" ACCUM                                               \
    "s_mov_b64 exec, s[12:13]\n v_cmpx_lt_f32_e32 s21, v12\n s_cbranch_execz 9f\n 1:\n"
#define STEP_L_TODAY                                                                                                   \
    "ds_read_b128 v[4:7], v40\n ds_read_b32 v3, v40 offset:16\n s_waitcnt lgkmcnt(1)\n"                                \
    "v_sub_f32_e32 v24, v4, v10\n v_sub_f32_e32 v25, v5, v11\n v_mul_f32_e32 v26, v24, v6\n s_waitcnt lgkmcnt(0)\n v_mul_f32_e32 v23, v25, v3\n" \
    "v_fmac_f32_e32 v26, v7, v25\n v_mul_f32_e32 v23, v25, v23\n v_fmac_f32_e32 v23, v26, v24\n"                       \
    "v_cmp_le_f32_e32 vcc, s20, v23\n s_and_b64 s[10:11], vcc, s[8:9]\n s_cmp_eq_u64 s[10:11], 0\n"                   \
    "s_cbranch_scc1 1f\n s_and_saveexec_b64 s[10:11], s[8:9]\n s_cbranch_execz 2f\n v_cmp_le_f32_e32 vcc, s20, v23\n ds_read_b128 v[30:33], v40 offset:32\n" \
    EXP2_POLY "s_waitcnt lgkmcnt(0)\n v_mul_f32_e32 v0, v33, v0\n v_cndmask_b32_e32 v0, 0, v0, vcc\n" ACCUM            \
    "2:\n s_or_b64 exec, exec, s[10:11]\n v_cmp_lt_f32_e64 s[8:9], s21, v12\n 1:\n"                                    \
    "s_cmp_eq_u64 s[8:9], 0\n s_cbranch_scc1 9f\n"

// M: operand banks of a three-source VALU (a VGPR's bank = its number mod 4).  Eight v_fmac per step, accumulators of bank 0.
#define FMAC8(A, B) "v_fmac_f32_e32 v20, " A ", " B "\n v_fmac_f32_e32 v24, " A ", " B "\n v_fmac_f32_e32 v28, " A ", " B "\n v_fmac_f32_e32 v32, " A ", " B "\n" \
                    "v_fmac_f32_e32 v36, " A ", " B "\n v_fmac_f32_e32 v40, " A ", " B "\n v_fmac_f32_e32 v44, " A ", " B "\n v_fmac_f32_e32 v48, " A ", " B "\n"
#define STEP_M_DISTINCT FMAC8("v5", "v10")   /* banks D 0, A 1, B 2 */
#define STEP_M_AB_SAME FMAC8("v6", "v10")    /* D 0, A 2, B 2 */
#define STEP_M_DA_SAME FMAC8("v4", "v10")    /* D 0, A 0, B 2 */
#define STEP_M_ALL_SAME FMAC8("v4", "v12")   /* D 0, A 0, B 0 */
#define MUL8(A, B) "v_mul_f32_e32 v20, " A ", " B "\n v_mul_f32_e32 v24, " A ", " B "\n v_mul_f32_e32 v28, " A ", " B "\n v_mul_f32_e32 v32, " A ", " B "\n" \
                   "v_mul_f32_e32 v36, " A ", " B "\n v_mul_f32_e32 v40, " A ", " B "\n v_mul_f32_e32 v44, " A ", " B "\n v_mul_f32_e32 v48, " A ", " B "\n"
#define STEP_M_MUL_DISTINCT MUL8("v5", "v10")
#define STEP_M_MUL_SAME MUL8("v6", "v10")
// N: the blend step (form d) with every three-source instruction's operands in three banks: dx v8 (0), dy v9 (1),
// a1 v14 (2), y v11 (3); weight v19 (3) against colours v4..v6 (0, 1, 2); sums v21, v22, v23 (1, 2, 3), t v12 (0)
#define STEP_N                                                                                                         \
    "v_sub_f32_e32 v8, v4, v10\n v_sub_f32_e32 v9, v5, v13\n v_mul_f32_e32 v14, v6, v8\n v_mul_f32_e32 v11, v3, v9\n"   \
    "v_fmac_f32_e32 v14, v7, v9\n v_mul_f32_e32 v11, v11, v9\n v_fmac_f32_e32 v11, v14, v8\n"                          \
    "v_cmp_le_f32_e32 vcc, s20, v11\n s_cbranch_vccz 1f\n s_and_saveexec_b64 s[12:13], vcc\n"                          \
    "v_min_f32 v0, 0x42fc0000, v11\n v_add_f32_e32 v1, 0x4b400000, v0\n v_add_f32_e32 v15, 0xcb400000, v1\n"           \
    "v_sub_f32_e32 v0, v0, v15\n v_fmamk_f32 v15, v0, 0x3aaddd0c, v16\n v_fmaak_f32 v15, v15, v0, 0x3d635ba9\n"         \
    "v_fmaak_f32 v15, v15, v0, 0x3e75fcde\n v_fmaak_f32 v15, v15, v0, 0x3f317215\n v_fma_f32 v0, v15, v0, 1.0\n"        \
    "v_lshl_add_u32 v0, v1, 23, v0\n v_mul_f32_e32 v19, v33, v0\n v_mul_f32_e32 v19, v19, v12\n"                        \
    "v_fmac_f32_e32 v21, v4, v19\n v_fmac_f32_e32 v22, v5, v19\n v_fmac_f32_e32 v23, v6, v19\n v_sub_f32_e32 v12, v12, v19\n" \
    "s_mov_b64 exec, s[12:13]\n v_cmpx_lt_f32_e32 s21, v12\n s_cbranch_execz 9f\n 1:\n"

// O: the exponent of FOUR splats at once on the idle matrix pipe (round 4's review, item 5): six v_mfma_f32_4x4x1_16b_f32,
// one per monomial (1, u, v, u^2, uv, v^2) of the lane's pixel about the quadrant origin (B operand: v50..v55, constant per
// lane) against the four splats' expanded coefficients (A operand: lane 4b + i carries splat i's coefficient, v56..v61 —
// in a real loop six ds_read_b32 per four splats), accumulated in order: bit for bit an fmaf chain.  D = four exponents per
// lane (v[40:43] / v[44:47]).  Software-pipelined as a real loop would be: the six MFMAs of the NEXT four splats are issued
// first, then the VALU half (cutoff test, exp2, blend: form d without its 7-instruction exponent) of the CURRENT four runs
// beside them.  One STEP_O = 8 splat-steps (set A then set B).
#define MFMA6(ACC)                                                                                                     \
    "v_mfma_f32_4x4x1_16b_f32 " ACC ", v56, v50, 0\n v_mfma_f32_4x4x1_16b_f32 " ACC ", v57, v51, " ACC "\n"          \
    "v_mfma_f32_4x4x1_16b_f32 " ACC ", v58, v52, " ACC "\n v_mfma_f32_4x4x1_16b_f32 " ACC ", v59, v53, " ACC "\n"     \
    "v_mfma_f32_4x4x1_16b_f32 " ACC ", v60, v54, " ACC "\n v_mfma_f32_4x4x1_16b_f32 " ACC ", v61, v55, " ACC "\n"
#define EXP2_POLY_OF(Y)                                                                                                \
    "v_min_f32 v0, 0x42fc0000, " Y "\n v_add_f32_e32 v1, 0x4b400000, v0\n v_add_f32_e32 v8, 0xcb400000, v1\n"         \
    "v_sub_f32_e32 v0, v0, v8\n v_fmamk_f32 v8, v0, 0x3aaddd0c, v14\n v_fmaak_f32 v8, v8, v0, 0x3d635ba9\n"            \
    "v_fmaak_f32 v8, v8, v0, 0x3e75fcde\n v_fmaak_f32 v8, v8, v0, 0x3f317215\n v_fma_f32 v0, v8, v0, 1.0\n"            \
    "v_lshl_add_u32 v0, v1, 23, v0\n"
#define TAIL_OF(Y, L)                                                                                                  \
    "v_cmp_le_f32_e32 vcc, s20, " Y "\n s_cbranch_vccz " L "f\n s_and_saveexec_b64 s[12:13], vcc\n"                   \
    EXP2_POLY_OF(Y) "v_mul_f32_e32 v0, v33, v0\n" ACCUM3                                                               \
    "s_mov_b64 exec, s[12:13]\n v_cmpx_lt_f32_e32 s21, v12\n s_cbranch_execz 9f\n " L ":\n"
#define STEP_O                                                                                                         \
    MFMA6("v[44:47]") TAIL_OF("v40", "1") TAIL_OF("v41", "2") TAIL_OF("v42", "3") TAIL_OF("v43", "4")                  \
    MFMA6("v[40:43]") TAIL_OF("v44", "5") TAIL_OF("v45", "6") TAIL_OF("v46", "7") TAIL_OF("v47", "1")
// P: the VALU half alone (what is left of a step once the exponent comes from somewhere else), 8 per STEP like O
#define STEP_P                                                                                                         \
    TAIL_OF("v40", "1") TAIL_OF("v41", "2") TAIL_OF("v42", "3") TAIL_OF("v43", "4")                                    \
    TAIL_OF("v44", "5") TAIL_OF("v45", "6") TAIL_OF("v46", "7") TAIL_OF("v47", "1")
// Q: O with the coefficient reads a real loop needs: six ds_read_b32 per four splats, per-lane addresses (four distinct)
#define LDS6 "ds_read_b32 v56, v62\n ds_read_b32 v57, v62 offset:4\n ds_read_b32 v58, v62 offset:8\n ds_read_b32 v59, v62 offset:12\n ds_read_b32 v60, v62 offset:16\n ds_read_b32 v61, v62 offset:20\n"
#define STEP_Q                                                                                                         \
    "s_waitcnt lgkmcnt(0)\n" MFMA6("v[44:47]") LDS6 TAIL_OF("v40", "1") TAIL_OF("v41", "2") TAIL_OF("v42", "3") TAIL_OF("v43", "4") \
    "s_waitcnt lgkmcnt(0)\n" MFMA6("v[40:43]") LDS6 TAIL_OF("v44", "5") TAIL_OF("v45", "6") TAIL_OF("v46", "7") TAIL_OF("v47", "1")
#define SETUP_O                                                                                                        \
    "v_and_b32 v50, 7, v62\n v_cvt_f32_u32 v51, v50\n v_lshrrev_b32 v52, 3, v62\n v_cvt_f32_u32 v52, v52\n"             \
    "v_mov_b32 v50, 1.0\n v_mul_f32 v53, v51, v51\n v_mul_f32 v54, v51, v52\n v_mul_f32 v55, v52, v52\n"               \
    "v_mov_b32 v56, 0xc0a00000\n v_mov_b32 v57, 0x3dcccccd\n v_mov_b32 v58, 0x3d4ccccd\n v_mov_b32 v59, 0xbc23d70a\n"  \
    "v_mov_b32 v60, 0x3a83126f\n v_mov_b32 v61, 0xbc23d70a\n"                                                          \
    "v_mov_b32 v40, 0xc0000000\n v_mov_b32 v41, 0xc0400000\n v_mov_b32 v42, 0xc0800000\n v_mov_b32 v43, 0xc0a00000\n"  \
    "v_mov_b32 v44, 0xc0000000\n v_mov_b32 v45, 0xc0400000\n v_mov_b32 v46, 0xc0800000\n v_mov_b32 v47, 0xc0a00000\n"

#define CLOBBERS "v0", "v1", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v19", "v20", "v21", "v22", \
                 "v23", "v24", "v25", "v26", "v28", "v30", "v31", "v32", "v33", "v36", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "s8", "s9", "s10", "s11", "s12", "s13", "s20", "s21", "s22", "vcc", "scc", "memory"

#define STEP_KERNEL(NAME, STEP, LDS)                                                                                   \
    __global__ void NAME(Out *out, float seed) {                                                                       \
        extern __shared__ float rec[];                                                                                 \
        if (LDS) {                                                                                                     \
            for (int i = threadIdx.x; i < 4096; i += blockDim.x) {                                                     \
                const int f = i % 12;                                                                                  \
                rec[i] = f == 0 ? 10.0f : f == 1 ? 11.0f : f == 2 ? -0.01f : f == 3 ? 0.001f : f == 4 ? -0.01f : f == 8 ? 0.5f : f == 9 ? 0.25f : f == 10 ? 0.125f : f == 11 ? 1e-6f : 0.0f; \
            }                                                                                                          \
            __syncthreads();                                                                                           \
        }                                                                                                              \
        unsigned long long t0, t1; float sink;                                                                         \
        const unsigned lds_addr = (unsigned)(size_t)rec + (threadIdx.x >> 6) * 48;                                     \
        asm volatile("v_mov_b32 v62, %4\n" SETUP_O "v_mov_b32 v62, %6\n v_mov_b32 v40, %3\n v_mov_b32 v0, %4\n" SETUP                                                    \
                     "s_memtime %0\n s_waitcnt lgkmcnt(0)\n s_movk_i32 s22, %5\n"                                      \
                     "8:\n" STEP STEP STEP STEP STEP STEP STEP STEP                                                    \
                     "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 8b\n"                               \
                     "9:\n s_mov_b64 exec, -1\n s_memtime %1\n v_add_f32 %2, v20, v12\n s_waitcnt lgkmcnt(0)\n"         \
                     : "=&s"(t0), "=&s"(t1), "=v"(sink) : "v"(lds_addr), "v"(threadIdx.x & 63), "n"(TRIPS), "v"((unsigned)(size_t)rec + (threadIdx.x & 3) * 48) : CLOBBERS); \
        if ((threadIdx.x & 63) == 0) { Out o; o.t0 = t0; o.t1 = t1; o.sink = sink + seed; out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = o; } \
    }

STEP_KERNEL(step_a_today, STEP_A, 0)
STEP_KERNEL(step_b_exec_mask, STEP_B, 0)
STEP_KERNEL(step_c_exec_alive, STEP_C, 0)
STEP_KERNEL(step_d_three_fmac, STEP_D, 0)
STEP_KERNEL(step_e_cut_today, STEP_E_TODAY, 0)
STEP_KERNEL(step_e_cut_exec, STEP_E_EXEC, 0)
STEP_KERNEL(step_f_exponent, STEP_F, 0)
STEP_KERNEL(step_g_exp2, STEP_G, 0)
STEP_KERNEL(step_h_accum, STEP_H, 0)
STEP_KERNEL(step_i_cndmask_vcc_x4, STEP_I_VCC, 0)
STEP_KERNEL(step_i_cndmask_sgpr_x4, STEP_I_SGPR, 0)
STEP_KERNEL(step_i_cndmask_vv_x4, STEP_I_VV, 0)
STEP_KERNEL(step_j_mul_sgpr_x4, STEP_J_S, 0)
STEP_KERNEL(step_j_mul_vgpr_x4, STEP_J_V, 0)
STEP_KERNEL(step_j_cmp_sgpr_x4, STEP_J_CMP_S, 0)
STEP_KERNEL(step_j_cmp_vgpr_x4, STEP_J_CMP_V, 0)
STEP_KERNEL(step_j_cmpx_x4, STEP_J_CMPX, 0)
STEP_KERNEL(step_l_lds_exec_alive, STEP_L, 1)
STEP_KERNEL(step_l_lds_today, STEP_L_TODAY, 1)
STEP_KERNEL(step_m_fmac_distinct_x8, STEP_M_DISTINCT, 0)
STEP_KERNEL(step_m_fmac_ab_same_x8, STEP_M_AB_SAME, 0)
STEP_KERNEL(step_m_fmac_da_same_x8, STEP_M_DA_SAME, 0)
STEP_KERNEL(step_m_fmac_all_same_x8, STEP_M_ALL_SAME, 0)
STEP_KERNEL(step_m_mul_distinct_x8, STEP_M_MUL_DISTINCT, 0)
STEP_KERNEL(step_m_mul_same_x8, STEP_M_MUL_SAME, 0)
STEP_KERNEL(step_n_banked, STEP_N, 0)
STEP_KERNEL(step_o_mfma_exponent_x8, STEP_O, 0)
STEP_KERNEL(step_p_valu_half_x8, STEP_P, 0)
STEP_KERNEL(step_q_mfma_exponent_lds_x8, STEP_Q, 1)

// Whole-launch throughput at full occupancy: 4 x 2048 workgroups of 4 waves (8 workgroups = 8 waves per SIMD fit a CU:
// 41 VGPRs, LDS below 20 KiB), every wave long enough (TRIPS) that dispatch does not matter; cycles per step per SIMD
// = wall time x 2.4 GHz x 1024 SIMDs / (waves x steps per wave).  The median wave's own s_memtime ticks per step / 8 is
// printed beside it (equal when 8 waves were resident per SIMD throughout and the clock is 2.4 GHz).
struct Rate { double per_simd, per_wave; };
template <typename K>
static Rate run(K kernel, Out *d_out, std::vector<Out> &h) {
    const int wgs = 2048 * 4;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    Rate best{1e30, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 16384, 0, d_out, 1.0f);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), d_out, sizeof(Out) * wgs * 4, hipMemcpyDeviceToHost));
        std::vector<double> c(wgs * 4);
        for (int i = 0; i < wgs * 4; ++i) c[i] = (double)(h[i].t1 - h[i].t0);
        std::nth_element(c.begin(), c.begin() + c.size() / 2, c.end());
        const double steps = (double)wgs * 4 * TRIPS * 8;
        Rate r{ms * 1e-3 * 2.4e9 * 1024.0 / steps, c[c.size() / 2] / (TRIPS * 8.0) / 8.0};
        if (r.per_simd < best.per_simd) best = r;
    }
    return best;
}

int main() {
    Out *d_out; CHECK(hipMalloc(&d_out, sizeof(Out) * 2048 * 4 * 4));
    std::vector<Out> h(2048 * 4 * 4);
    printf("| sequence | instructions (VALU + SALU/branch + LDS) | cycles per step per SIMD, 8 waves per SIMD (wall clock at 2.4 GHz) | median wave's ticks per step / 8 |\n|---|---|---|---|\n");
#define ROW(NAME, WHAT) { Rate a = run(NAME, d_out, h); printf("| %s | %s | %.1f | %.1f |\n", #NAME, WHAT, a.per_simd, a.per_wave); fflush(stdout); }
    ROW(step_a_today, "27 + 9 + 0: the seen step hipcc emits today, record in registers")
    ROW(step_b_exec_mask, "25 + 6 + 0: below-cutoff lanes masked through exec, no v_cndmask, no second compare")
    ROW(step_c_exec_alive, "25 + 5 + 0: exec is the alive mask, v_cmpx retires pixels")
    ROW(step_d_three_fmac, "26 + 5 + 0: as c with three v_fmac instead of v_pk_fma + v_fmac")
    ROW(step_e_cut_today, "8 + 6 + 0: a step that ends at the cutoff test, today")
    ROW(step_e_cut_exec, "8 + 2 + 0: ... with exec as the alive mask")
    ROW(step_f_exponent, "7: the quadratic form")
    ROW(step_g_exp2, "10: exp2 by the contract's polynomial")
    ROW(step_h_accum, "4: weight, three channels, transmittance")
    ROW(step_i_cndmask_vcc_x4, "4 v_cndmask_b32_e32 (mask in vcc, src0 = 0)")
    ROW(step_i_cndmask_sgpr_x4, "4 v_cndmask_b32_e64 (mask in s[8:9])")
    ROW(step_i_cndmask_vv_x4, "4 v_cndmask_b32_e32 (mask in vcc, two VGPR sources)")
    ROW(step_j_mul_sgpr_x4, "4 v_mul_f32 with an SGPR operand")
    ROW(step_j_mul_vgpr_x4, "4 v_mul_f32, VGPR operands")
    ROW(step_j_cmp_sgpr_x4, "4 v_cmp_le_f32 vcc, SGPR, VGPR")
    ROW(step_j_cmp_vgpr_x4, "4 v_cmp_le_f32 vcc, VGPR, VGPR")
    ROW(step_j_cmpx_x4, "4 v_cmpx_lt_f32")
    ROW(step_l_lds_exec_alive, "step c with the record read from LDS (b128 + b32, then b128)")
    ROW(step_l_lds_today, "step a with the record read from LDS")
    ROW(step_m_fmac_distinct_x8, "8 v_fmac, operands in three banks (D 0, A 1, B 2)")
    ROW(step_m_fmac_ab_same_x8, "8 v_fmac, the two multiplicands in one bank (D 0, A 2, B 2)")
    ROW(step_m_fmac_da_same_x8, "8 v_fmac, accumulator and a multiplicand in one bank (D 0, A 0, B 2)")
    ROW(step_m_fmac_all_same_x8, "8 v_fmac, all three in one bank")
    ROW(step_m_mul_distinct_x8, "8 v_mul, sources in two banks")
    ROW(step_m_mul_same_x8, "8 v_mul, sources in one bank")
    ROW(step_n_banked, "26 + 5 + 0: form d with every three-source instruction's operands in three banks")
    // (one STEP of the three rows below is EIGHT splat-steps: divide their cycles by 8 to compare with form d)
    ROW(step_o_mfma_exponent_x8, "8 x (19 + 4) VALU/branch + 12 MFMA: exponents of 4 splats by six v_mfma_f32_4x4x1 (pipelined), rest of form d on the VALU; PER 8 STEPS")
    ROW(step_p_valu_half_x8, "8 x (19 + 4): the VALU half alone (form d without its exponent); PER 8 STEPS")
    ROW(step_q_mfma_exponent_lds_x8, "as o, plus the 12 ds_read_b32 of the coefficients (per-lane addresses, four distinct); PER 8 STEPS")
    return 0;
}
