// Device helpers shared by the decode kernels (skinny.hip, attention.hip, declayer.hip): int8 -> fp16 expansion, the
// instruction-lean LayerNorm arithmetic of the fused GEMV prologues, DPP reductions, the 8-element fp16 dot product
// and the k-blocked activation index.  Header-only; every function is static __forceinline__.
#pragma once
#include "common.h"

// 8 weight bytes (q + 128) -> 8 exact fp16 integers: v_perm_b32 drops each byte into the mantissa of
// 0x6400 (= 1024.0, whose ulp is 1), a packed subtract of 1152 leaves q.  4 + 4 VALU per 8 weights.
static __device__ __forceinline__ half8 q8_to_half8(uint2 v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 off = {(h16)1152.f, (h16)1152.f};
    const unsigned w[2] = {v.x, v.y};
    half8 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned lo = __builtin_amdgcn_perm(0x64646464u, w[i], 0x04010400u);
        const unsigned hi = __builtin_amdgcn_perm(0x64646464u, w[i], 0x04030402u);
        const h2 a = __builtin_bit_cast(h2, lo) - off, b = __builtin_bit_cast(h2, hi) - off;
        r[4 * i + 0] = a[0]; r[4 * i + 1] = a[1]; r[4 * i + 2] = b[0]; r[4 * i + 3] = b[1];
    }
    return r;
}

// Epilogue of a decode GEMV: accumulated tile value x row scale (int8 weights; 1 otherwise) + bias, as ONE fused
// multiply-add rounded to fp32 -- and KEPT in fp32 until the caller converts it.  Two things went wrong with less:
//  * left to the compiler (`t * s + b` under the default fp contraction) the copies of this epilogue in different kernels
//    came out as v_fma_f32 in one and v_mul + v_add in another;
//  * spelled fmaf(), the compiler still folded "fma, then convert to fp16" into v_fma_mixlo/hi_f16 -- ONE rounding,
//    straight to fp16 -- in some copies (the fused decode launch's GEMV role, two of four columns) and left
//    v_fmac_f32 + v_cvt_f16_f32 -- TWO roundings -- in others (skinny_kernel).  The results differ when the fp32
//    rounding crosses an fp16 tie: about one value in 30 000, with int8 weights only (with scale 1 the fma is an add and
//    is not folded).  That was the "last-bit difference of the int8 instance" of DESIGN 5c: found by dumping the epilogue
//    operands of a mismatching element from both kernels (the dump stores made it vanish: they keep the fp32 value alive)
//    and diffing the ISA of the builds with and without them (tools/dbg_q8.py).
// The empty asm pins the fp32 value in a register, so every copy rounds twice, like the stored-activation path.
// the same pin for any other fp32 value whose next stop is a conversion to fp16 in code that has a twin elsewhere
static __device__ __forceinline__ float pin_f32(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
static __device__ __forceinline__ float scale_bias(float t, float s, float b) {
    float r = __builtin_fmaf(t, s, b);
    asm volatile("" : "+v"(r));
    return r;
}

// ---- LayerNorm of a row slice held in registers (32 threads per row, NC chunks of 8 halves per thread).
// The prologue is VALU work every block repeats for all rows, so it is written for instruction count:
//   sums     v_dot2c_f32_f16 (x, 1) and (x, x): no conversions; variance = E[x^2] - mean^2 (fp32 sums of exact fp16
//            products; the rows are activations of O(1..10), far from cancellation at fp16 output precision)
//   apply    u = x * rstd - mean * rstd      v_fma_mix_f32   (fp16 source, fp32 result)
//            y = u * gamma + beta -> fp16    v_fma_mixlo/hi_f16 (fp16 gamma / beta sources, rounds once to fp16)
// = 2 VALU per element instead of ~6 (cvt, sub, mul, cvt, cvt, fma, cvt) plus 2 x 0.5 for the sums.
typedef _Float16 wx_h2v __attribute__((ext_vector_type(2)));
typedef unsigned wx_u4 __attribute__((ext_vector_type(4)));

template <int HI>
static __device__ __forceinline__ float mix_hff(unsigned xp, float a, float b) {      // (half HI of xp) * a + b
    float r;
    if (HI)
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(xp), "v"(a), "v"(b));
    else
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(xp), "v"(a), "v"(b));
    return r;
}
template <int HI>
static __device__ __forceinline__ void mix_fhh_to_h(unsigned& out, float u, unsigned gp, unsigned bp) {   // half HI of out = u * g + b
    if (HI)
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "+v"(out) : "v"(u), "v"(gp), "v"(bp));
    else
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(out) : "v"(u), "v"(gp), "v"(bp));
}

// sum over the 32 lanes that share a LayerNorm row: four DPP steps inside each 16-lane row (quad_perm, quad_perm,
// row_half_mirror, row_mirror) and one cross-row exchange; every lane ends with the total
static __device__ __forceinline__ float sum32_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    v += __shfl_xor(v, 16, 64);
    return v;
}

// sum over the 16 lanes of one DPP row (the four in-row steps of sum32_dpp)
static __device__ __forceinline__ float sum16_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

// partial (sum, sum of squares) of one chunk
static __device__ __forceinline__ void ln_accum(half8 x, float& s1, float& s2) {
    const wx_h2v one = {(h16)1.f, (h16)1.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const wx_h2v p = {x[j], x[j + 1]};
        s1 = __builtin_amdgcn_fdot2(p, one, s1, false);
        s2 = __builtin_amdgcn_fdot2(p, p, s2, false);
    }
}
// normalised chunk: ((x - mean) * rstd) * g + b with rstd_n = -mean * rstd
static __device__ __forceinline__ half8 ln_apply(half8 x, half8 g, half8 b, float rstd, float nmr) {
    const wx_u4 xu = __builtin_bit_cast(wx_u4, x), gu = __builtin_bit_cast(wx_u4, g), bu = __builtin_bit_cast(wx_u4, b);
    wx_u4 o = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        unsigned ow = 0u;
        mix_fhh_to_h<0>(ow, mix_hff<0>(xu[w], rstd, nmr), gu[w], bu[w]);
        mix_fhh_to_h<1>(ow, mix_hff<1>(xu[w], rstd, nmr), gu[w], bu[w]);
        o[w] = ow;
    }
    return __builtin_bit_cast(half8, o);
}

// Sum over the 8 lanes that share one key (lane & 7 = 16-byte chunk of the 64-wide head), on the DPP
// path of the VALU: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror.  (__shfl_xor compiles to
// ds_bpermute: 3 LDS round trips per key group in the hottest loop of the decode step.)
static __device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    return v;
}

typedef _Float16 wx_h2 __attribute__((ext_vector_type(2)));
// 8-element fp16 dot product with fp32 accumulation: 4 x v_dot2c_f32_f16, no conversions
static __device__ __forceinline__ float dot8_f16(half8 a, half8 b) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) acc = __builtin_amdgcn_fdot2((wx_h2){a[j], a[j + 1]}, (wx_h2){b[j], b[j + 1]}, acc, false);
    return acc;
}

// element index of (row b, feature n) of a decode activation [B][d]: row-major, or the k-blocked layout
// [n / 32][16 rows][32] that the following GEMV reads with contiguous fragment loads (<= 16 rows)
static __device__ __forceinline__ long act_index(int b, int n, long ld, int blocked) {
    // rows beyond 16 (coalesced requests): one blocked image of 16 * ld elements per group of 16 rows
    return blocked ? (long)(b >> 4) * 16 * ld + (long)(n >> 5) * 512 + (b & 15) * 32 + (n & 31) : (long)b * ld + n;
}

