// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere; fp16 storage, fp32 accumulation.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WX_WAVE 64

// Lab knobs read from the environment (tools/r05_*.sh: WX_DL_POLL, WX_NO_WIDE_GEMV, WX_WIDE_GEMV_FROM_17, WX_GEMM_4W,
// WX_GEMM_STAGGER_US, WX_GEMM_NT) exist in LAB builds only -- `python tools/build_lab.py env WX_LAB_ENV`; the shipped library
// reads no environment variable and compiles every one of them to its default.
#ifdef WX_LAB_ENV
#include <cstdlib>
#define WX_LAB_GETENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define WX_LAB_GETENV_INT(name, dflt) (dflt)
#endif

// erf GELU, as nn.GELU / mlx nn.gelu: 0.5 x (1 + erf(x / sqrt 2)) = 0.5 x + 0.5 |x| erf(|x| / sqrt 2).
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. below fp32 round-off of the result for
// every x where the fp16 output is not zero): one v_rcp, one v_exp and 8 FMAs, branch-free - the
// libm erff costs ~3x as many VALU slots, which is what the GEMM epilogues were spending.
__device__ __forceinline__ float gelu_f(float x) {
    const float az = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(t, q, 1.421413741f);
    q = fmaf(t, q, -0.284496736f);
    q = fmaf(t, q, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(az * az * -1.44269504088896341f);
    const float er = fmaf(-(t * q), e, 1.0f);
    return fmaf(0.5f * fabsf(x), er, 0.5f * x);
}

// Two GELUs side by side (the GEMM epilogues walk their accumulators in pairs).  Same arithmetic as gelu_f.
// Written on 2-vectors for the packed-fp32 VALU originally; the library is now built WITHOUT packed-fp32 ops
// (build.py: v_pk_*_f32 returns wrong bits in lanes 48-63 beside another wave's MFMAs), so this compiles to two
// interleaved scalar chains, which also hides the v_rcp / v_exp latency of one behind the other.
typedef float wx_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wx_f2 gelu_f2(wx_f2 x) {
    const wx_f2 ax = {fabsf(x[0]), fabsf(x[1])};
    const wx_f2 az = ax * 0.70710678118654752440f;
    const wx_f2 d = __builtin_elementwise_fma(az, (wx_f2){0.3275911f, 0.3275911f}, (wx_f2){1.0f, 1.0f});
    const wx_f2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    wx_f2 q = __builtin_elementwise_fma(t, (wx_f2){1.061405429f, 1.061405429f}, (wx_f2){-1.453152027f, -1.453152027f});
    q = __builtin_elementwise_fma(t, q, (wx_f2){1.421413741f, 1.421413741f});
    q = __builtin_elementwise_fma(t, q, (wx_f2){-0.284496736f, -0.284496736f});
    q = __builtin_elementwise_fma(t, q, (wx_f2){0.254829592f, 0.254829592f});
    const wx_f2 ea = az * az * -1.44269504088896341f;
    const wx_f2 e = {__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};
    const wx_f2 er = __builtin_elementwise_fma(-(t * q), e, (wx_f2){1.0f, 1.0f});
    return __builtin_elementwise_fma(ax * 0.5f, er, x * 0.5f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide reductions through a small LDS scratch (>= 32 floats); all threads get the result
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): consecutive
// logical tiles land on one XCD so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

#define WX_CHECK_HIP(expr)                                                        \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) return wx_fail(ctx, _e, #expr, __FILE__, __LINE__); \
    } while (0)
