// Skinny GEMMs (GEMVs) of the decode loop: a few activation rows against a [N][K] fp16 (or int8 + row scale)
// weight that is streamed from HBM exactly once per decode position.  Three kernels:
//   skinny_kernel      <= 16 rows.  One block = 16 / 8 / 4 output columns; its 8 (or 16) waves split K; every lane
//                      issues all of its 16-byte weight loads up front (straight to VGPRs: nothing is shared between
//                      waves), then feeds MFMA 16x16x32 f16 with the weight rows in the A role, so a lane ends up with
//                      4 consecutive output columns of one activation row.  Partial tiles are summed across the waves
//                      through LDS in a fixed order and wave 0 applies row scale / bias / GELU / residual.
//                      Optional fused LayerNorm: rows, gamma and beta are requested BEFORE the weights (vmcnt retires in
//                      issue order), gamma / beta travel once per block through LDS, statistics use v_dot2, the
//                      normalisation v_fma_mix (2 VALU per element), the normalised rows sit in LDS for the MFMAs.
//                      More than 16 rows (coalesced requests, <= 128): grid.y = group of 16 rows; the groups of a column
//                      tile re-read its weights from L2.
//   skinny_mt_kernel   decode step variant 3 (<= 64 rows): M-tiled, ceil(N / #CU) columns per block, one block per CU;
//                      every block reads ALL activation rows, which costs more than the row groups' weight re-reads.
//   skinny2_kernel     the 133 MB tied-embedding logits GEMV (no prologue; also split-K for decode step variant 2).
// All three keep the same per-element summation order, so a row decodes to the same tokens in any of them.
// No implicit fused multiply-adds in this file: `a * b + c` stays a multiply and an add unless the source says fmaf().
// The same expression is compiled several times here (template instances, roles inlined into different kernels), and
// under the default contraction the compiler fused it in one copy and not in another -- with int8 weights the fused
// launch's GEMV role, its self-computed-query path and skinny_kernel then disagreed in the last bit now and then, and a
// row's log-probability depended on which copy had produced its query.  Everything hot already spells its FMAs out.
// (The last of those differences was not a contraction at all: fma + conversion to fp16 folded into one v_fma_mix*_f16 in
// one copy -- decode_dev.h, scale_bias.)
#pragma clang fp contract(off)
#include "common.h"
#include "kernels.h"
#include "decode_dev.h"
#include <mutex>
#include <cstdlib>

#ifdef LAB_NO_W            // lab builds only (tools/build_lab.py): every weight load of the decode GEMVs hits one cached tile
#define WX_LAB_W(x) 0
#else
#define WX_LAB_W(x) (x)
#endif

namespace {

constexpr int SK_WAVES = 8;
constexpr int SK_MAXSTEPS = 20;   // k-steps (of 32) per wave: K <= 8*20*32 = 5120

template <bool LN, int STEPS, int WAVES, bool Q8 = false>
__global__ __launch_bounds__(64 * WAVES, (WAVES == 16 ? 4 : (STEPS <= 5 ? 4 : 2))) void skinny_kernel(SkinnyArgs p) {
#ifdef LAB_GEMV_PRIO      // lab: the GEMV waves at raised issue priority beside other passes' attention waves
    __builtin_amdgcn_s_setprio(3);
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);                  // [8][64][4] f32 = 8 KiB
    h16* a_lds = reinterpret_cast<h16*>(smem + WAVES * 64 * 16);  // LN: [16][K+8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tn = p.tile_n > 0 ? p.tile_n : 16;   // distinct output columns of this block (4, 8 or 16)
    const int n0 = blockIdx.x * tn;
    const int lda_s = p.K + 8;
    if (p.prof && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        // launch timer of the launch in front of this one (SkinnyArgs::prof): this launch's first block starts when that one
        // has drained (same stream), give or take the dispatch gap (0.04 - 1 us)
        const unsigned long long t0 = __hip_atomic_load(p.prof, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t0) {
            const unsigned long long now = (unsigned long long)__builtin_amdgcn_s_memrealtime();
            __hip_atomic_store(p.prof + 1, __hip_atomic_load(p.prof + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (now - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.prof + 2, __hip_atomic_load(p.prof + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.prof, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (gridDim.y > 1) {
        // more than 16 rows (coalesced requests): blockIdx.y = group of 16 rows.  The groups of one column tile read
        // the same weights; dispatched within microseconds of each other, all but the first find them in L2.
        const long g = blockIdx.y;
        p.A += g * 16 * (p.a_blocked ? (long)p.K : p.lda);          // k-blocked layout: 16 * K elements per group
        if (p.out_h) p.out_h += g * 16 * (p.out_blocked ? (long)p.N : p.ldo);
        if (p.out_f) p.out_f += g * 16 * p.ldo;
        if (p.R) p.R += g * 16 * p.ldr;
        p.M = min(16, p.M - 16 * (int)g);
    }

    // (0) LayerNorm input first: vmcnt retires in issue order, so the rows (L2 hits) must be ahead of
    //     the weight loads (HBM) in the queue or the statistics wait for the weights as well
    constexpr int MAXC = STEPS <= 5 ? 5 : 8;      // 32 threads x MAXC chunks x 8 = K <= 1280 / 2048
    half8 xv[LN ? MAXC : 1];
    if (LN) {
        const h16* xr0 = p.A + (long)min(tid >> 5, p.M - 1) * p.lda;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) xv[c] = *reinterpret_cast<const half8*>(xr0 + min((tid & 31) + 32 * c, (p.K >> 3) - 1) * 8);
    }
    // gamma / beta travel once per block, one 16-byte chunk per thread, still ahead of the weights, and are
    // shared through the (not yet used) partial-sum area of LDS: the normalisation then never waits for HBM
    half8 gbv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (LN) {
        const int nch0 = p.K >> 3;
        const int t2 = min(tid, 2 * nch0 - 1);
        gbv = *reinterpret_cast<const half8*>((t2 < nch0 ? p.ln_g : p.ln_b - (long)nch0 * 8) + (long)t2 * 8);
    }
    // (1) every weight load of this wave goes out next, unconditionally (clamped index, no
    //     branch): HBM latency is the longest pole, everything below overlaps with it
    const int nks = p.K >> 5;
    const int ks0 = (wave * nks) / WAVES, ks1 = ((wave + 1) * nks) / WAVES;
    const int nstep = ks1 - ks0;
    const int nrow = min(n0 + min(fr, tn - 1), p.N - 1);   // narrow tiles: surplus MFMA rows repeat the last one
    // weight row of this lane and the stride of a k-step: row-major [N][K], or the tile-blocked layout (SkinnyArgs::w_blocked)
    const long wrow = p.w_blocked ? ((long)(nrow >> 4) * (p.K >> 5) * 16 + (nrow & 15)) * 32 : (long)nrow * p.ldw;
    const int wks = p.w_blocked ? 512 : 32;
    half8 wreg[Q8 ? 1 : STEPS];
    uint2 wq[Q8 ? STEPS : 1];
    if (Q8) {
        const unsigned char* wp = p.Wq + WX_LAB_W(wrow) + fq * 8;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) wq[i] = *reinterpret_cast<const uint2*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * wks));
    } else {
        const h16* wp = p.W + WX_LAB_W(wrow) + fq * 8;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) wreg[i] = *reinterpret_cast<const half8*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * wks));
    }
    half8 areg[LN ? 1 : STEPS];
    if (!LN) {
        // row-major: 16 rows x 64 B per wave load (16 half-used lines); k-blocked: one contiguous KiB
        const h16* ap = p.a_blocked ? p.A + fr * 32 + fq * 8 : p.A + (long)min(fr, p.M - 1) * p.lda + fq * 8;
        const int kstride = p.a_blocked ? 512 : 32;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) areg[i] = *reinterpret_cast<const half8*>(ap + (long)min(ks0 + i, nks - 1) * kstride);
    }
    // (2) epilogue operands of wave 0, also up front: ONE 8-byte load each, from an address that is
    //     always valid (clamped, or the weight tile itself when there is no bias / residual).  Per-element
    //     conditional loads compile to eight serialized "load; s_waitcnt vmcnt(0)" round trips, each of
    //     which also waits for the weight loads above -- the whole block sat behind wave 0 for that long.
    const int em = fr, enb = n0 + 4 * fq;
    half4 eb4 = {0, 0, 0, 0}, er4 = {0, 0, 0, 0};
    // aligned fast path: 4 / 8 / 16 columns per block; any other tile_n (e.g. N / #CU for one balanced round of blocks)
    // takes per-column operands -- still requested here, unconditionally, from clamped addresses
    const bool tile4 = (tn & 3) == 0;
    const bool evec = tile4 && enb + 3 < p.N;
    const int ncol = max(0, min(4, min(tn - 4 * fq, p.N - enb)));     // valid columns of this lane
    if (wave == 0) {
        const h16* dummy = p.A;      // always a valid address (>= 8 bytes)
        if (tile4) {
            const int nc = min(enb, (p.N - 4) & ~3);
            eb4 = *reinterpret_cast<const half4*>(p.bias ? p.bias + nc : dummy);
            er4 = *reinterpret_cast<const half4*>(p.R ? p.R + (long)min(em, p.M - 1) * p.ldr + nc : dummy);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = min(enb + r, p.N - 1);
                eb4[r] = p.bias ? p.bias[n] : dummy[0];
                er4[r] = p.R ? p.R[(long)min(em, p.M - 1) * p.ldr + n] : dummy[0];
            }
        }
    }
    f32x4 es4 = {1.f, 1.f, 1.f, 1.f};
    if (Q8 && wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) es4[r] = p.wscale[min(enb + r, p.N - 1)];
    }

    if (LN) {
        // (3) LayerNorm of the 16 rows from registers: 32 threads per row (ln_accum / ln_apply above)
        const int row = tid >> 5, sub = tid & 31;
        const int nch = p.K >> 3;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (sub + 32 * c < nch) ln_accum(xv[c], s1, s2);
        s1 = sum32_dpp(s1);
        s2 = sum32_dpp(s2);
        const float mean = s1 / (float)p.K;
        const float rstd = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
        const float nmr = -mean * rstd;
        h16* gb_lds = reinterpret_cast<h16*>(smem);          // [2][K] halves <= 8 KiB: the partial-sum area, free until the MFMAs are done
        if (tid < 2 * nch) *reinterpret_cast<half8*>(gb_lds + tid * 8) = gbv;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = sub + 32 * c;
            if (ch < nch) {
                const half8 g = *reinterpret_cast<const half8*>(gb_lds + ch * 8);
                const half8 be = *reinterpret_cast<const half8*>(gb_lds + (nch + ch) * 8);
                *reinterpret_cast<half8*>(a_lds + row * lda_s + ch * 8) = ln_apply(xv[c], g, be, rstd, nmr);
            }
        }
        __syncthreads();
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        if (i < nstep) {
            half8 af;
            if (LN)
                af = *reinterpret_cast<const half8*>(a_lds + fr * lda_s + (ks0 + i) * 32 + fq * 8);
            else
                af = areg[LN ? 0 : i];
            const half8 wf = Q8 ? q8_to_half8(wq[Q8 ? i : 0]) : wreg[Q8 ? 0 : i];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af, acc, 0, 0, 0);
        }
    }
    *reinterpret_cast<f32x4*>(part + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wave == 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (w * 64 + lane) * 4);
            t += v;
        }
        // lane: activation row m = fr, output columns n = n0 + 4*fq + r
        if (em < p.M && ncol > 0) {
            float v[4];
            if (evec || !tile4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = scale_bias(t[r], es4[r], p.bias ? (float)eb4[r] : 0.f);
                    if (p.gelu) v[r] = gelu_f(v[r]);
                    v[r] = pin_f32(v[r] + (p.R ? (float)er4[r] : 0.f));     // (fp32 sum with an fp16 operand, then fp16: foldable -- decode_dev.h)
                }
            }
#if defined(LAB_DUMP_Q8) && !defined(LAB_DUMP_Q8_NOSTORES)
            if (p.lab_dump && enb == 608) {
                float* dd = p.lab_dump + (p.lab_slot * 16 + em) * 8;
                dd[0] = t[0]; dd[1] = es4[0]; dd[2] = (float)eb4[0]; dd[3] = v[0]; dd[4] = t[1]; dd[5] = es4[1]; dd[6] = (float)eb4[1]; dd[7] = v[1];
            }
#endif
            if (evec) {
                if (p.out_f) {
                    *reinterpret_cast<f32x4*>(p.out_f + (long)em * p.ldo + enb) = (f32x4){v[0], v[1], v[2], v[3]};
                } else {
                    half4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                    const long oaddr = p.out_blocked ? (long)(enb >> 5) * 512 + em * 32 + (enb & 31) : (long)em * p.ldo + enb;
                    *reinterpret_cast<half4*>(p.out_h + oaddr) = o;
                }
            } else if (!tile4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < ncol) {
                        if (p.out_f)
                            p.out_f[(long)em * p.ldo + enb + r] = v[r];
                        else
                            p.out_h[(long)em * p.ldo + enb + r] = (h16)v[r];
                    }
            } else {
                for (int r = 0; r < 4 && enb + r < p.N; ++r) {   // ragged last tile of an N that is not a multiple of 4
                    float x = scale_bias(t[r], es4[r], p.bias ? (float)p.bias[enb + r] : 0.f);
                    if (p.gelu) x = gelu_f(x);
                    if (p.R) x += (float)p.R[(long)em * p.ldr + enb + r];
                    if (p.out_f)
                        p.out_f[(long)em * p.ldo + enb + r] = x;
                    else
                        p.out_h[(long)em * p.ldo + enb + r] = (h16)x;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// The K = 4d GEMV (FC2) for launches that share the GPU with other passes: the arithmetic of skinny_kernel<false, 10, 16>
// -- sixteen k-slices, partial tiles summed in slice order: the same bits -- on EIGHT waves, each taking two slices one
// after the other.  A 1024-thread block needs a whole CU's worth of free registers at once; beside another pass's
// cross-attention blocks (two 512-thread blocks of 128 registers per CU) it waits for a CU to drain, and with three
// passes in flight FC2 took 81 us a launch against 20 alone (profiles/r03_bench_overlap.txt).  512-thread blocks of under
// 128 registers slot in beside them.
template <bool Q8>
__global__ __launch_bounds__(512, 4) void skinny_vw2_kernel(SkinnyArgs p) {
#ifdef LAB_GEMV_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    constexpr int STEPS = 10, WAVES = 8, VWAVES = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);                  // [16][64][4] f32 = 16 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int tn = p.tile_n > 0 ? p.tile_n : 16;
    const int n0 = blockIdx.x * tn;
    if (gridDim.y > 1) {
        const long g = blockIdx.y;
        p.A += g * 16 * (p.a_blocked ? (long)p.K : p.lda);
        if (p.out_h) p.out_h += g * 16 * (p.out_blocked ? (long)p.N : p.ldo);
        if (p.out_f) p.out_f += g * 16 * p.ldo;
        if (p.R) p.R += g * 16 * p.ldr;
        p.M = min(16, p.M - 16 * (int)g);
    }
    const int nks = p.K >> 5;
    const int nrow = min(n0 + min(fr, tn - 1), p.N - 1);
    // weight row of this lane and the stride of a k-step: row-major [N][K], or the tile-blocked layout (SkinnyArgs::w_blocked)
    const long wrow = p.w_blocked ? ((long)(nrow >> 4) * (p.K >> 5) * 16 + (nrow & 15)) * 32 : (long)nrow * p.ldw;
    const int wks = p.w_blocked ? 512 : 32;
    const h16* ap = p.a_blocked ? p.A + fr * 32 + fq * 8 : p.A + (long)min(fr, p.M - 1) * p.lda + fq * 8;
    const int kstride = p.a_blocked ? 512 : 32;
    // epilogue operands of wave 0 first (one 8-byte load each from an always valid address, see skinny_kernel)
    const int em = fr, enb = n0 + 4 * fq;
    half4 eb4 = {0, 0, 0, 0}, er4 = {0, 0, 0, 0};
    const bool tile4 = (tn & 3) == 0;
    const bool evec = tile4 && enb + 3 < p.N;
    const int ncol = max(0, min(4, min(tn - 4 * fq, p.N - enb)));
    f32x4 es4 = {1.f, 1.f, 1.f, 1.f};
    if (wave == 0) {
        const h16* dummy = p.A;
        if (tile4) {
            const int nc = min(enb, (p.N - 4) & ~3);
            eb4 = *reinterpret_cast<const half4*>(p.bias ? p.bias + nc : dummy);
            er4 = *reinterpret_cast<const half4*>(p.R ? p.R + (long)min(em, p.M - 1) * p.ldr + nc : dummy);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = min(enb + r, p.N - 1);
                eb4[r] = p.bias ? p.bias[n] : dummy[0];
                er4[r] = p.R ? p.R[(long)min(em, p.M - 1) * p.ldr + n] : dummy[0];
            }
        }
        if (Q8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) es4[r] = p.wscale[min(enb + r, p.N - 1)];
        }
    }
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const int vw = 2 * wave + ph;                                  // the k-slice of wave `vw` of the 16-wave kernel
        const int ks0 = (vw * nks) / VWAVES, ks1 = ((vw + 1) * nks) / VWAVES;
        const int nstep = ks1 - ks0;
        half8 wreg[Q8 ? 1 : STEPS];
        uint2 wq[Q8 ? STEPS : 1];
        if (Q8) {
            const unsigned char* wp = p.Wq + WX_LAB_W(wrow) + fq * 8;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) wq[i] = *reinterpret_cast<const uint2*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * wks));
        } else {
            const h16* wp = p.W + WX_LAB_W(wrow) + fq * 8;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) wreg[i] = *reinterpret_cast<const half8*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * wks));
        }
        half8 areg[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) areg[i] = *reinterpret_cast<const half8*>(ap + (long)min(ks0 + i, nks - 1) * kstride);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            if (i < nstep) {
                const half8 wf = Q8 ? q8_to_half8(wq[Q8 ? i : 0]) : wreg[Q8 ? 0 : i];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, areg[i], acc, 0, 0, 0);
            }
        }
        *reinterpret_cast<f32x4*>(part + (vw * 64 + lane) * 4) = acc;
    }
    __syncthreads();
    if (wave == 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < VWAVES; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (w * 64 + lane) * 4);
            t += v;
        }
        if (em < p.M && ncol > 0) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = scale_bias(t[r], es4[r], p.bias ? (float)eb4[r] : 0.f);
                if (p.gelu) v[r] = gelu_f(v[r]);
                v[r] = pin_f32(v[r] + (p.R ? (float)er4[r] : 0.f));
            }
            if (evec) {
                if (p.out_f) {
                    *reinterpret_cast<f32x4*>(p.out_f + (long)em * p.ldo + enb) = (f32x4){v[0], v[1], v[2], v[3]};
                } else {
                    half4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                    const long oaddr = p.out_blocked ? (long)(enb >> 5) * 512 + em * 32 + (enb & 31) : (long)em * p.ldo + enb;
                    *reinterpret_cast<half4*>(p.out_h + oaddr) = o;
                }
            } else {
                for (int r = 0; r < ncol; ++r) {
                    if (p.out_f)
                        p.out_f[(long)em * p.ldo + enb + r] = v[r];
                    else
                        p.out_h[(long)em * p.ldo + enb + r] = (h16)v[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// More than 16 rows in ONE pass over the weights (round 5).  skinny_kernel takes a launch of 17..128 rows as grid.y =
// groups of 16 rows: every group is a block of its own that loads the column tile's weights again (from L2), so a
// 112-row FC1 is 2 240 blocks -- three rounds of resident blocks, each a chain of load -> (LayerNorm) -> MFMA ->
// cross-wave sum -> store -- and takes 43 us alone for 13 MB of weights.  Here a block keeps its weight fragments in
// registers and walks FOUR row groups with them: the same eight (or sixteen) k-slices per row, the same MFMA order
// inside a slice, the partial tiles summed in slice order -- a row gets the bits skinny_kernel / skinny_vw2_kernel give
// it -- with the activation fragments (k-blocked layout: one contiguous KiB per group and k-step, L2 hits) requested one
// k-step ahead of their MFMAs.  A quarter of the blocks, a quarter of the weight reads, and each block's chain is
// paid once for four groups.  A fused LayerNorm is not part of it: ln_rows32_blk_kernel normalises the rows first, with
// skinny_kernel's prologue arithmetic (32 threads per row, the same chunk order, dot2 statistics, fma_mix), into the
// k-blocked layout this kernel reads.
constexpr int SW_GB = 4;          // row groups (of 16 rows) per block

template <int NSL, int STEPS, bool Q8>
__global__ __launch_bounds__(512, 2) void skinny_wide_kernel(SkinnyArgs p) {
    constexpr int PH = NSL / 8;     // k-slices per wave, taken one after the other (skinny_vw2_kernel's order for 16 slices)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);                  // [NSL][SW_GB][64][4] f32
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int tn = p.tile_n > 0 ? p.tile_n : 16;
    const int n0 = blockIdx.x * tn;
    const int G = (p.M + 15) >> 4, g0 = blockIdx.y * SW_GB, ng = min(SW_GB, G - g0);
    if (p.prof && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        // launch timer of the launch in front of this one (SkinnyArgs::prof), as in skinny_kernel
        const unsigned long long t0 = __hip_atomic_load(p.prof, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t0) {
            const unsigned long long now = (unsigned long long)__builtin_amdgcn_s_memrealtime();
            __hip_atomic_store(p.prof + 1, __hip_atomic_load(p.prof + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (now - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.prof + 2, __hip_atomic_load(p.prof + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.prof, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const int nks = p.K >> 5;
    const int nrow = min(n0 + min(fr, tn - 1), p.N - 1);
    const long wrow = p.w_blocked ? ((long)(nrow >> 4) * (p.K >> 5) * 16 + (nrow & 15)) * 32 : (long)nrow * p.ldw;
    const int wks = p.w_blocked ? 512 : 32;
    // activation fragment of (group g, k-step ks): k-blocked [g][ks][16 rows][32], or row-major rows of the group
    const long gstride = 16 * (p.a_blocked ? (long)p.K : p.lda);
    const int kstride = p.a_blocked ? 512 : 32;
    const h16* ap[SW_GB];
#pragma unroll
    for (int g = 0; g < SW_GB; ++g) {
        const int gg = g0 + min(g, ng - 1);                     // absent groups repeat the last one (never stored)
        const int mg = min(16, p.M - 16 * gg);
        ap[g] = p.A + gg * gstride + (p.a_blocked ? fr * 32 + fq * 8 : (long)min(fr, mg - 1) * p.lda + fq * 8);
    }
    // epilogue operands of the waves that will finish a group (wave g finishes group g0 + g): requested first, from
    // addresses that are always valid (see skinny_kernel)
    const int em = fr, enb = n0 + 4 * fq;
    const int myg = g0 + min((int)wave, ng - 1);
    const int m_g = min(16, p.M - 16 * myg);
    half4 eb4 = {0, 0, 0, 0}, er4 = {0, 0, 0, 0};
    f32x4 es4 = {1.f, 1.f, 1.f, 1.f};
    if (wave < SW_GB) {
        const h16* dummy = p.A;
        const int nc = min(enb, (p.N - 4) & ~3);
        eb4 = *reinterpret_cast<const half4*>(p.bias ? p.bias + nc : dummy);
        er4 = *reinterpret_cast<const half4*>(p.R ? p.R + ((long)myg * 16 + min(em, m_g - 1)) * p.ldr + nc : dummy);
        if (Q8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) es4[r] = p.wscale[min(enb + r, p.N - 1)];
        }
    }
#pragma unroll 1
    for (int ph = 0; ph < PH; ++ph) {
        const int vw = PH * wave + ph;                             // the k-slice (wave `vw` of the NSL-wave kernel)
        const int ks0 = (vw * nks) / NSL, ks1 = ((vw + 1) * nks) / NSL;
        const int nstep = ks1 - ks0;
        half8 wreg[Q8 ? 1 : STEPS];
        uint2 wq[Q8 ? STEPS : 1];
        if (Q8) {
            const unsigned char* wp = p.Wq + WX_LAB_W(wrow) + fq * 8;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) wq[i] = *reinterpret_cast<const uint2*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * wks));
        } else {
            const h16* wp = p.W + WX_LAB_W(wrow) + fq * 8;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) wreg[i] = *reinterpret_cast<const half8*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * wks));
        }
        f32x4 acc[SW_GB];
#pragma unroll
        for (int g = 0; g < SW_GB; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (STEPS <= 5) {
            // every activation fragment of the slice is requested up front, behind the weights (80 registers at 5 k-steps x 4
            // groups): the L2 round trips overlap the weights' HBM round trip instead of following it one k-step at a time
            half8 areg[STEPS][SW_GB];
#pragma unroll
            for (int i = 0; i < STEPS; ++i)
#pragma unroll
                for (int g = 0; g < SW_GB; ++g) areg[i][g] = *reinterpret_cast<const half8*>(ap[g] + (long)min(ks0 + i, nks - 1) * kstride);
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                if (i < nstep) {
                    const half8 wf = Q8 ? q8_to_half8(wq[Q8 ? i : 0]) : wreg[Q8 ? 0 : i];
#pragma unroll
                    for (int g = 0; g < SW_GB; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, areg[i][g], acc[g], 0, 0, 0);
                }
            }
        } else {
            // ten k-steps: the fragments of all of them do not fit the registers beside the weights; one k-step ahead
            half8 acur[SW_GB], anxt[SW_GB];
#pragma unroll
            for (int g = 0; g < SW_GB; ++g) acur[g] = *reinterpret_cast<const half8*>(ap[g] + (long)min(ks0, nks - 1) * kstride);
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                if (i + 1 < STEPS) {
#pragma unroll
                    for (int g = 0; g < SW_GB; ++g) anxt[g] = *reinterpret_cast<const half8*>(ap[g] + (long)min(ks0 + i + 1, nks - 1) * kstride);
                }
                if (i < nstep) {
                    const half8 wf = Q8 ? q8_to_half8(wq[Q8 ? i : 0]) : wreg[Q8 ? 0 : i];
#pragma unroll
                    for (int g = 0; g < SW_GB; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, acur[g], acc[g], 0, 0, 0);
                }
                if (i + 1 < STEPS) {
#pragma unroll
                    for (int g = 0; g < SW_GB; ++g) acur[g] = anxt[g];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < SW_GB; ++g) *reinterpret_cast<f32x4*>(part + ((vw * SW_GB + g) * 64 + lane) * 4) = acc[g];
    }
    __syncthreads();
    if (wave < ng) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NSL; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + ((w * SW_GB + wave) * 64 + lane) * 4);
            t += v;
        }
        // lane: activation row m = fr of group g0 + wave, output columns n = n0 + 4*fq + r
        if (em < m_g && enb + 3 < p.N && 4 * fq < tn) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = scale_bias(t[r], es4[r], p.bias ? (float)eb4[r] : 0.f);
                if (p.gelu) v[r] = gelu_f(v[r]);
                v[r] = pin_f32(v[r] + (p.R ? (float)er4[r] : 0.f));
            }
            const long grow = (long)(g0 + wave) * 16;
            if (p.out_f) {
                *reinterpret_cast<f32x4*>(p.out_f + (grow + em) * p.ldo + enb) = (f32x4){v[0], v[1], v[2], v[3]};
            } else {
                half4 o = {(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3]};
                const long oaddr = p.out_blocked ? grow * p.N + (long)(enb >> 5) * 512 + em * 32 + (enb & 31) : (grow + em) * p.ldo + enb;
                *reinterpret_cast<half4*>(p.out_h + oaddr) = o;
            }
        }
    }
}

// LayerNorm of the rows of a wide launch for skinny_wide_kernel: skinny_kernel's fused prologue as a launch of its own --
// one block = one group of 16 rows, 32 threads per row, chunks sub, sub + 32, ... of 8 elements, ln_accum in that order,
// sum32_dpp, ln_apply -- written in the k-blocked layout [group][k / 32][16 rows][32].  MAXC = 5 (K <= 1280) / 8 (<= 2048)
// as the instances of skinny_kernel have it.
template <int MAXC>
__global__ __launch_bounds__(512) void ln_rows32_blk_kernel(const h16* __restrict__ x, long ldx, const h16* __restrict__ gam,
                                                            const h16* __restrict__ bet, h16* __restrict__ y, int M, int K) {
    const int tid = threadIdx.x, g = blockIdx.x, row = tid >> 5, sub = tid & 31, nch = K >> 3;
    const int mg = min(16, M - 16 * g);
    const h16* xr0 = x + ((long)g * 16 + min(row, mg - 1)) * ldx;
    half8 xv[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) xv[c] = *reinterpret_cast<const half8*>(xr0 + min(sub + 32 * c, nch - 1) * 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (sub + 32 * c < nch) ln_accum(xv[c], s1, s2);
    s1 = sum32_dpp(s1);
    s2 = sum32_dpp(s2);
    const float mean = s1 / (float)K;
    const float rstd = rsqrtf(fmaxf(s2 / (float)K - mean * mean, 0.f) + 1e-5f);
    const float nmr = -mean * rstd;
    h16* yg = y + (long)g * 16 * K;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int ch = sub + 32 * c;
        if (ch < nch) {
            const half8 g8 = *reinterpret_cast<const half8*>(gam + ch * 8);
            const half8 b8 = *reinterpret_cast<const half8*>(bet + ch * 8);
            *reinterpret_cast<half8*>(yg + (long)(ch >> 2) * 512 + row * 32 + (ch & 3) * 8) = ln_apply(xv[c], g8, b8, rstd, nmr);
        }
    }
}

// ---------------------------------------------------------------------------------------
// M-tiled, column-balanced variant: up to 64 activation rows (MT tiles of 16) against a
// tile of `tile_n` <= 32 weight rows (NT MFMA tiles), ONE block per CU.  The host picks
// tile_n = ceil(N / #CU) so every GEMV of the decode step is exactly one balanced round of blocks
// (N = d -> 5 columns, 3d -> 15, 4d -> 20 on 256 CUs; with 16-column tiles N = 4d is 320 blocks =
// 1.25 rounds and N = d only 80-160 CUs pull on HBM).  Each weight fragment is loaded once and
// used for all MT row tiles, so coalescing several 16-chunk requests into one decode launch
// reads the weights once instead of once per request.
template <bool LN, int STEPS, int MT, int NT, bool Q8 = false>
__global__ __launch_bounds__(512, 1) void skinny_mt_kernel(SkinnyArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16* a_lds = reinterpret_cast<h16*>(smem);          // LN: [16*MT][K+8] fp16
    float* part = reinterpret_cast<float*>(smem);       // after the MFMAs: [8][MT*NT][64][4] f32
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int tn = p.tile_n;
    const int n0 = blockIdx.x * tn;
    const int lda_s = p.K + 8;
    const int nks = p.K >> 5;
    const int ks0 = (wave * nks) / 8, ks1 = ((wave + 1) * nks) / 8;
    const int nstep = ks1 - ks0;

    // Load order = dependency order (vmcnt retires in issue order): LayerNorm rows of ALL row tiles and gamma / beta
    // first (L2 hits), then every weight fragment (HBM).  One block per CU, so there are registers to hold them all.
    constexpr int MAXC = 5;      // 32 threads x 5 chunks x 8 = K <= 1280
    const int sub = tid & 31, nch = p.K >> 3;
    half8 xv[LN ? MT : 1][LN ? MAXC : 1], gv[LN ? MAXC : 1], bv[LN ? MAXC : 1];
    if (LN) {
#pragma unroll
        for (int g = 0; g < MT; ++g) {
            const h16* xr = p.A + (long)min(g * 16 + (tid >> 5), p.M - 1) * p.lda;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) xv[g][c] = *reinterpret_cast<const half8*>(xr + min(sub + 32 * c, nch - 1) * 8);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = min(sub + 32 * c, nch - 1);
            gv[c] = *reinterpret_cast<const half8*>(p.ln_g + ch * 8);
            bv[c] = *reinterpret_cast<const half8*>(p.ln_b + ch * 8);
        }
    }
    half8 wreg[Q8 ? 1 : STEPS][NT];
    uint2 wq[Q8 ? STEPS : 1][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int nrow = min(n0 + min(16 * nt + fr, tn - 1), p.N - 1);
        if (Q8) {
            const unsigned char* wp = p.Wq + WX_LAB_W((long)nrow * p.ldw) + fq * 8;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) wq[i][nt] = *reinterpret_cast<const uint2*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * 32));
        } else {
            const h16* wp = p.W + WX_LAB_W((long)nrow * p.ldw) + fq * 8;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) wreg[i][nt] = *reinterpret_cast<const half8*>(wp + WX_LAB_W(min(ks0 + i, nks - 1) * 32));
        }
    }

    // epilogue operands of the waves that will finish a tile (wave w < MT*NT owns tile (w / NT, w % NT)): all eight
    // 2-byte loads go out now, from clamped (always valid) addresses -- per-element conditional loads in the epilogue
    // compile to serialized load / s_waitcnt pairs at the very end of the kernel
    float eb[4] = {0.f, 0.f, 0.f, 0.f}, er[4] = {0.f, 0.f, 0.f, 0.f}, es[4] = {1.f, 1.f, 1.f, 1.f};
    if (wave < MT * NT) {
        const int emt = wave / NT, ent = wave - emt * NT;
        const int erow = min(emt * 16 + fr, p.M - 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = min(n0 + 16 * ent + 4 * fq + r, p.N - 1);
            const h16* bsrc = p.bias ? p.bias + n : p.A;
            const h16* rsrc = p.R ? p.R + (long)erow * p.ldr + n : p.A;
            eb[r] = (float)*bsrc;
            er[r] = (float)*rsrc;
            if (Q8) es[r] = p.wscale[n];
        }
    }

    if (LN) {
#pragma unroll
        for (int g = 0; g < MT; ++g) {
            const int row = g * 16 + (tid >> 5);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (sub + 32 * c < nch) ln_accum(xv[g][c], s1, s2);
            s1 = sum32_dpp(s1);
            s2 = sum32_dpp(s2);
            const float mean = s1 / (float)p.K;
            const float rstd = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = sub + 32 * c;
                if (ch < nch) *reinterpret_cast<half8*>(a_lds + row * lda_s + ch * 8) = ln_apply(xv[g][c], gv[c], bv[c], rstd, nmr);
            }
        }
        __syncthreads();
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const h16* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = p.A + (long)min(mt * 16 + fr, p.M - 1) * p.lda + fq * 8;
    // activations straight from L2 (no LN): software-pipelined in groups of G k-steps, one group of loads in flight
    // under the previous group's MFMAs -- a load per step inside the loop is STEPS dependent L2 round trips
    constexpr int G = STEPS <= 5 ? 5 : (NT == 1 ? (MT <= 3 ? 5 : 4) : (MT == 1 ? 5 : MT == 2 ? 3 : MT == 3 ? 2 : 1));   // register budget
    constexpr int NG = (STEPS + G - 1) / G;
    half8 abuf[2][LN ? 1 : G][LN ? 1 : MT];
    auto load_group = [&](int gi, int slot) {
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                abuf[slot][LN ? 0 : j][LN ? 0 : mt] = *reinterpret_cast<const half8*>(ap[mt] + min(ks0 + gi * G + j, nks - 1) * 32);
    };
    if (!LN) load_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        if (!LN && gi + 1 < NG) load_group(gi + 1, (gi + 1) & 1);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int i = gi * G + j;
            if (i < STEPS && i < nstep) {
                half8 af[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (LN)
                        af[mt] = *reinterpret_cast<const half8*>(a_lds + (mt * 16 + fr) * lda_s + (ks0 + i) * 32 + fq * 8);
                    else
                        af[mt] = abuf[gi & 1][LN ? 0 : j][LN ? 0 : mt];
                }
                half8 wf[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wf[nt] = Q8 ? q8_to_half8(wq[Q8 ? i : 0][nt]) : wreg[Q8 ? 0 : i][nt];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    if (LN) __syncthreads();     // a_lds is dead from here on: its space carries the partial tiles
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<f32x4*>(part + (((wave * MT + mt) * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
    __syncthreads();
    if (wave >= MT * NT) return;
    const int mt = wave / NT, nt = wave - mt * NT;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) t += *reinterpret_cast<const f32x4*>(part + (((w * MT + mt) * NT + nt) * 64 + lane) * 4);
    const int em = mt * 16 + fr;
    if (em >= p.M) return;
    const int nlim = min(n0 + tn, p.N);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * nt + 4 * fq + r;
        if (n < nlim) {
            float v = scale_bias(t[r], Q8 ? es[r] : 1.f, p.bias ? eb[r] : 0.f);
            if (p.gelu) v = gelu_f(v);
            if (p.R) v += er[r];
            if (p.out_f)
                p.out_f[(long)em * p.ldo + n] = v;
            else
                p.out_h[(long)em * p.ldo + n] = (h16)v;
        }
    }
}

template <auto KERN>
static void launch_mt_kernel(const SkinnyArgs& a, int nb, size_t lds, hipStream_t s) {
    // the dynamic LDS of the LN variants can exceed the 64 KiB default: raise the limit once per kernel
    static std::once_flag raised;      // per kernel instantiation; host threads of different engine contexts may race here
    if (lds > 64 * 1024)
        std::call_once(raised, [] { (void)hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
    hipLaunchKernelGGL(KERN, dim3(nb), dim3(512), lds, s, a);
}

template <bool LN, int STEPS, int MT>
hipError_t launch_skinny_mt_nt(const SkinnyArgs& a, int nb, int nt, size_t lds, hipStream_t s) {
    if (a.Wq) {
        if (nt == 1) launch_mt_kernel<skinny_mt_kernel<LN, STEPS, MT, 1, true>>(a, nb, lds, s);
        else launch_mt_kernel<skinny_mt_kernel<LN, STEPS, MT, 2, true>>(a, nb, lds, s);
    } else {
        if (nt == 1) launch_mt_kernel<skinny_mt_kernel<LN, STEPS, MT, 1, false>>(a, nb, lds, s);
        else launch_mt_kernel<skinny_mt_kernel<LN, STEPS, MT, 2, false>>(a, nb, lds, s);
    }
    return hipGetLastError();
}

template <bool LN, int STEPS>
hipError_t launch_skinny_mt_m(const SkinnyArgs& a, int nb, int mt, int nt, size_t lds, hipStream_t s) {
    switch (mt) {
        case 1: return launch_skinny_mt_nt<LN, STEPS, 1>(a, nb, nt, lds, s);
        case 2: return launch_skinny_mt_nt<LN, STEPS, 2>(a, nb, nt, lds, s);
        case 3: return launch_skinny_mt_nt<LN, STEPS, 3>(a, nb, nt, lds, s);
        default: return launch_skinny_mt_nt<LN, STEPS, 4>(a, nb, nt, lds, s);
    }
}

}  // namespace

hipError_t launch_skinny_mt(const SkinnyArgs& a0, int n_cu, hipStream_t s) {
    if (a0.w_blocked) return hipErrorInvalidValue;     // the M-tiled kernel reads row-major weights
    SkinnyArgs a = a0;
    if ((a.K & 31) || a.K > 8 * 20 * 32 || a.M < 1 || a.M > 64 || (a.ln_g && a.K > 1280)) return hipErrorInvalidValue;
    int tn = (a.N + n_cu - 1) / n_cu;
    if (tn > 32) tn = 32;
    a.tile_n = tn;
    const int nb = (a.N + tn - 1) / tn, nt = (tn + 15) / 16, mt = (a.M + 15) / 16;
    const int steps = ((a.K >> 5) + 7) / 8;
    const size_t lds_part = (size_t)8 * mt * nt * 1024;
    if (a.ln_g) {
        size_t lds = (size_t)16 * mt * (a.K + 8) * 2;
        if (lds < lds_part) lds = lds_part;
        if (lds > 150 * 1024) return hipErrorInvalidValue;   // 64 rows x K = 1280 does not fit the CU's LDS: <= 48 rows there
        return launch_skinny_mt_m<true, 5>(a, nb, mt, nt, lds, s);
    }
    if (steps <= 5) return launch_skinny_mt_m<false, 5>(a, nb, mt, nt, lds_part, s);
    return launch_skinny_mt_m<false, 20>(a, nb, mt, nt, lds_part, s);
}

template <bool LN, int STEPS, int WAVES>
static void launch_v1(const SkinnyArgs& a, int nb, size_t lds, hipStream_t s) {
    const dim3 grid(nb, (a.M + 15) / 16);
    if (a.Wq)
        hipLaunchKernelGGL((skinny_kernel<LN, STEPS, WAVES, true>), grid, dim3(64 * WAVES), lds, s, a);
    else
        hipLaunchKernelGGL((skinny_kernel<LN, STEPS, WAVES, false>), grid, dim3(64 * WAVES), lds, s, a);
}

hipError_t launch_skinny(const SkinnyArgs& a, hipStream_t s) {
    if ((a.K & 31) || a.K > SK_WAVES * SK_MAXSTEPS * 32 || a.M < 1 || a.M > 128) return hipErrorInvalidValue;
    if (a.M > 16 && a.out_blocked && a.ldo != a.N) return hipErrorInvalidValue;   // one blocked image of 16 * N per row group
    const int tn = a.tile_n > 0 ? a.tile_n : 16;
    if (tn < 1 || tn > 16) return hipErrorInvalidValue;
    if (a.out_blocked && ((tn & 3) || (a.N & 31) || a.out_f)) return hipErrorInvalidValue;
    if (a.a_blocked && a.ln_g) return hipErrorInvalidValue;
    if (a.w_blocked && ((a.N & 15) || (a.K & 31))) return hipErrorInvalidValue;
    const int nb = (a.N + tn - 1) / tn;
    const int steps = ((a.K >> 5) + SK_WAVES - 1) / SK_WAVES;
    if (a.ln_g && (a.K > 2048 || (steps <= 5 && a.K > 1280))) return hipErrorInvalidValue;
    // more than 16 rows: one pass over the weights for four row groups (skinny_wide_kernel) where the caller asks for it
    // (no_wide: -1 from 17 rows on, 0 from 65 rows on, 1 never) and its slicing is the one the 16-row launch of the same GEMV
    // uses (8 slices of <= 10 k-steps, or the K = 4d kernel's 16 slices of <= 10).
    // Measured (profiles/r05_ab_wide_*.txt, r05_kernel_stats_112x1_*.csv): ALONE the row-group kernels are the faster ones --
    // 112 rows: 14.4 us per LN + GEMV and 7.7 us per projection against 10.2 us + a 5.0 us LayerNorm launch --; with three
    // passes in flight the one-pass kernels win from four row groups on (64 rows +2.1 %, the driver's 112 + 112 + 96 +3.2 %:
    // a quarter of the blocks compete with the other passes' attention blocks for wave slots, whose launch runs 229 us
    // instead of 253) and lose below (48 rows -2.1 %, 32 rows -8.8 %).  The decode step decides (api.hip, decode_step_v1).
    const bool sixteen = !a.ln_g && steps > 10 && a.wide_block;
    if (a.M > (a.no_wide < 0 ? 16 : 64) && a.no_wide <= 0 && (tn & 3) == 0 && a.N % tn == 0 && (a.N & 3) == 0 && (steps <= 10 || sixteen) &&
        (!a.ln_g || a.ln_scratch) && !(a.out_blocked && a.out_f)) {
        SkinnyArgs w = a;
        if (a.ln_g) {
            const int G = (a.M + 15) / 16;
            if (steps <= 5)
                hipLaunchKernelGGL(ln_rows32_blk_kernel<5>, dim3(G), dim3(512), 0, s, a.A, a.lda, a.ln_g, a.ln_b, a.ln_scratch, a.M, a.K);
            else
                hipLaunchKernelGGL(ln_rows32_blk_kernel<8>, dim3(G), dim3(512), 0, s, a.A, a.lda, a.ln_g, a.ln_b, a.ln_scratch, a.M, a.K);
            w.A = a.ln_scratch; w.lda = a.K; w.a_blocked = 1; w.ln_g = nullptr; w.ln_b = nullptr;
        }
        const dim3 grid(nb, ((a.M + 15) / 16 + SW_GB - 1) / SW_GB);
        const size_t lds = (size_t)(sixteen ? 16 : 8) * SW_GB * 64 * 16;
#define WX_WIDE(NSL, ST)                                                                                              \
        do {                                                                                                          \
            static std::once_flag once_;                                                                              \
            std::call_once(once_, [] {                                                                                \
                (void)hipFuncSetAttribute((const void*)skinny_wide_kernel<NSL, ST, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * SW_GB * 64 * 16); \
                (void)hipFuncSetAttribute((const void*)skinny_wide_kernel<NSL, ST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * SW_GB * 64 * 16);  \
            });                                                                                                       \
            if (w.Wq)                                                                                                 \
                hipLaunchKernelGGL((skinny_wide_kernel<NSL, ST, true>), grid, dim3(512), lds, s, w);                  \
            else                                                                                                      \
                hipLaunchKernelGGL((skinny_wide_kernel<NSL, ST, false>), grid, dim3(512), lds, s, w);                 \
        } while (0)
        if (sixteen) WX_WIDE(16, 10);
        else if (steps <= 2) WX_WIDE(8, 2);
        else if (steps <= 5) WX_WIDE(8, 5);
        else WX_WIDE(8, 10);
#undef WX_WIDE
        return hipGetLastError();
    }
    if (a.ln_g) {
        const size_t lds = SK_WAVES * 64 * 16 + (size_t)16 * (a.K + 8) * 2;
        if (steps <= 2)
            launch_v1<true, 2, 8>(a, nb, lds, s);
        else if (steps <= 5)
            launch_v1<true, 5, 8>(a, nb, lds, s);
        else
            launch_v1<true, 8, 8>(a, nb, lds, s);
    } else {
        const size_t lds = SK_WAVES * 64 * 16;
        if (steps <= 2)
            launch_v1<false, 2, 8>(a, nb, lds, s);
        else if (steps <= 5)
            launch_v1<false, 5, 8>(a, nb, lds, s);
        else if (steps <= 10)
            launch_v1<false, 10, 8>(a, nb, lds, s);
        else if (a.wide_block && a.M > 16 && (a.tile_n & 3) == 0) {
            // K = 4d, wide launches (passes in flight beside other passes): the same sixteen k-slices on eight waves
            const dim3 grid(nb, (a.M + 15) / 16);
            if (a.Wq)
                hipLaunchKernelGGL(skinny_vw2_kernel<true>, grid, dim3(512), 16 * 64 * 16, s, a);
            else
                hipLaunchKernelGGL(skinny_vw2_kernel<false>, grid, dim3(512), 16 * 64 * 16, s, a);
        } else if (a.wide_block)   // K = 4d: 16 waves x 10 k-steps keep twice as many requests in flight per CU
            launch_v1<false, 10, 16>(a, nb, 16 * 64 * 16, s);
        else
            launch_v1<false, 20, 8>(a, nb, lds, s);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// v2: split-K across BLOCKS.  The decode GEMVs are a few MB each; with one block per 16
// output columns only N/16 (= 80 for N = d) compute units ever pull on HBM and each block
// is a chain of dependent phases, so the v1 kernel sat at ~1 TB/s.  Here the grid is
// (N/16, ksplit): every block streams a (16 x K/ksplit) weight slab with all of its loads
// in flight at once (weights and the matching activation fragments straight to VGPRs, no
// LDS staging, no LayerNorm prologue -- the normalised activations are produced once per
// step by resln_kernel).  ksplit == 1 blocks apply the epilogue; ksplit > 1 blocks write
// fp32 partial tiles that the consumer (resln_kernel / attention prologue) sums in a fixed
// order, so results are bitwise reproducible (no float atomics).
namespace {

constexpr int S2_WAVES = 4;
constexpr int S2_MAXSTEPS = 10;   // k-steps per wave: K / ksplit <= 4 * 10 * 32 = 1280

template <int MT>
__global__ __launch_bounds__(256) void skinny2_kernel(Skinny2Args p) {
    __shared__ __attribute__((aligned(16))) float part[S2_WAVES * MT * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, ky = blockIdx.y;
    const int nks_total = p.K >> 5;
    const int kb0 = (ky * nks_total) / p.ksplit, kb1 = ((ky + 1) * nks_total) / p.ksplit;
    const int nks = kb1 - kb0;
    const int ks0 = kb0 + (wave * nks) / S2_WAVES, ks1 = kb0 + ((wave + 1) * nks) / S2_WAVES;
    const int nrow = min(n0 + fr, p.N - 1);
    const h16* wp = p.W + WX_LAB_W((long)nrow * p.ldw) + fq * 8;

    half8 wreg[S2_MAXSTEPS], areg[S2_MAXSTEPS][MT];
#pragma unroll
    for (int i = 0; i < S2_MAXSTEPS; ++i)
        if (ks0 + i < ks1) wreg[i] = *reinterpret_cast<const half8*>(wp + (ks0 + i) * 32);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const h16* ap = p.A + (long)min(mt * 16 + fr, p.M - 1) * p.lda + fq * 8;
#pragma unroll
        for (int i = 0; i < S2_MAXSTEPS; ++i)
            if (ks0 + i < ks1) areg[i][mt] = *reinterpret_cast<const half8*>(ap + (ks0 + i) * 32);
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < S2_MAXSTEPS; ++i)
        if (ks0 + i < ks1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[i], areg[i][mt], acc[mt], 0, 0, 0);
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(part + ((wave * MT + mt) * 64 + lane) * 4) = acc[mt];
    __syncthreads();
    if (wave >= MT) return;
    const int mt = wave;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < S2_WAVES; ++w) t += *reinterpret_cast<const f32x4*>(part + ((w * MT + mt) * 64 + lane) * 4);
    const int m = mt * 16 + fr;
    if (m >= p.M) return;
    const int nb = n0 + 4 * fq;
    if (p.ksplit > 1) {
        float* dst = p.part + ((long)ky * 16 + m) * p.ldp + nb;
        if (nb + 3 < p.N) {
            *reinterpret_cast<f32x4*>(dst) = t;
        } else {
            for (int r = 0; r < 4 && nb + r < p.N; ++r) dst[r] = t[r];
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        if (n < p.N) {
            float v = t[r];
            if (p.bias) v += (float)p.bias[n];
            if (p.gelu) v = gelu_f(v);
            if (p.out_f)
                p.out_f[(long)m * p.ldo + n] = v;
            else
                p.out_h[(long)m * p.ldo + n] = (h16)v;
        }
    }
}

// The same GEMV for N >> #CU (the 133 MB tied-embedding logits): a block walks several 16-column tiles (tile = block,
// block + grid, ...).  Its waves load their K-slice of the activations ONCE (with one tile per block the activation
// fragments were re-read from L2 for every tile: as many bytes again as the weights), and the next tile's weight
// loads are issued as soon as the MFMAs have consumed the registers, so they fly during this tile's cross-wave sum
// and stores.  Per-element summation order is that of skinny2_kernel (same k-slices per wave, waves summed 0..3).
// LN (MT = 1): the block normalises the 16 rows itself (16 threads per row, the decode GEMVs' dot2 / fma_mix LayerNorm
// arithmetic), parks them in LDS and every wave takes its K-slice from there -- the separate final-LayerNorm launch of
// the decode step is gone; the first tile's weights are requested before the prologue and fly under it.
constexpr int S2P_ROW = 1280 * 2 + 16;   // LDS row stride of the normalised activations (K <= 1280, +16 B against bank aliasing)
// RH = 2 (more than 32 rows): a second group of four waves takes rows 32..63 of the SAME tiles with the same four k-slices
// -- its weight loads ask for the addresses the first group has just asked for and are served by the CU's own cache, so
// the 133 MB still cross the fabric once, and every row keeps its summation order.  (Four row tiles in one group of waves
// is 232 registers: skinny2_kernel<4>, 81 us a launch at 64 rows against 27 at 16.)
template <int MT, bool LN, int RH = 1>
__global__ __launch_bounds__(256 * RH) void skinny2p_kernel(Skinny2Args p, int ntiles) {
    __shared__ __attribute__((aligned(16))) float part_all[2][RH][S2_WAVES * MT * 64 * 4];
    __shared__ __attribute__((aligned(16))) char xn_l[LN ? 16 * S2P_ROW : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & (S2_WAVES - 1), rh = tid >> 8;     // k-slice, row half
    const int fr = lane & 15, fq = lane >> 4;
    const int nks_total = p.K >> 5;
    const int ks0 = (wave * nks_total) / S2_WAVES, ks1 = ((wave + 1) * nks_total) / S2_WAVES;
    static_assert(S2_WAVES == 4, "row half = tid >> 8");

    half8 wreg[S2_MAXSTEPS], areg[S2_MAXSTEPS][MT];
    int tile = blockIdx.x;
    auto load_w = [&](int tl) {
        const int nrow = min(tl * 16 + fr, p.N - 1);
        const h16* wp = p.W + WX_LAB_W((long)nrow * p.ldw) + fq * 8;
        // unconditional (clamped) loads: with the loads behind per-step branches the compiler cannot count what is in
        // flight and falls back to waiting for almost everything before the first use of anything
#pragma unroll
        for (int i = 0; i < S2_MAXSTEPS; ++i) wreg[i] = *reinterpret_cast<const half8*>(wp + max(0, min(ks0 + i, ks1 - 1)) * 32);
    };
    if constexpr (LN) {
        static_assert(MT == 1 && RH == 1, "fused LayerNorm: one row tile");
        // thread (row = tid / 16, t16 = tid % 16) owns chunks t16, t16 + 16, ... of its row (8 halves each).
        // Request order (vmcnt retires in issue order): rows, first half of gamma / beta, THEN the first tile's weights
        // -- the statistics and half of the normalisation run while the weights are still on their way from HBM.
        const int row = tid >> 4, t16 = tid & 15, nch = p.K >> 3;
        const h16* xr = p.A + (long)min(row, p.M - 1) * p.lda;
        half8 xv[10], gb[2];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = min(t16 + 16 * i, nch - 1);
            xv[i] = *reinterpret_cast<const half8*>(xr + c * 8);
        }
        // gamma | beta: 2 * nch chunks (<= 5 KB), fetched once per block (<= 2 per thread) and shared through the two
        // (still idle, 8 KB) partial-sum buffers
        static_assert(sizeof(part_all) >= 2 * (1280 / 8) * 16, "gamma | beta do not fit in the partial-sum buffers");
        char* gb_l = reinterpret_cast<char*>(part_all);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = min(tid + 256 * j, 2 * nch - 1);
            gb[j] = *reinterpret_cast<const half8*>((idx < nch ? p.ln_g + idx * 8 : p.ln_b + (idx - nch) * 8));
        }
        __builtin_amdgcn_sched_barrier(0);
        load_w(min(tile, ntiles - 1));   // unconditional (grid <= ntiles): a conditional load cannot be counted by the compiler
        __builtin_amdgcn_sched_barrier(0);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
            ln_accum((t16 + 16 * i < nch) ? xv[i] : zero8, s1, s2);
        }
        s1 = sum16_dpp(s1);
        s2 = sum16_dpp(s2);
        const float mean = s1 / (float)p.K;
        const float rstd = rsqrtf(fmaxf(s2 / (float)p.K - mean * mean, 0.f) + 1e-5f);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<half8*>(gb_l + min(tid + 256 * j, 2 * nch - 1) * 16) = gb[j];   // clamped duplicates write the same value
        // LDS-only barriers here: __syncthreads() would also wait for the weight loads in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int c = t16 + 16 * i;
            if (c < nch) {
                const half8 g8 = *reinterpret_cast<const half8*>(gb_l + c * 16);
                const half8 b8 = *reinterpret_cast<const half8*>(gb_l + (nch + c) * 16);
                *reinterpret_cast<half8*>(xn_l + row * S2P_ROW + c * 16) = ln_apply(xv[i], g8, b8, rstd, nmr);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < S2_MAXSTEPS; ++i)
            areg[i][0] = *reinterpret_cast<const half8*>(xn_l + fr * S2P_ROW + (max(0, min(ks0 + i, ks1 - 1)) * 32 + fq * 8) * 2);
    } else {
        load_w(min(tile, ntiles - 1));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h16* ap = p.A + (long)min((rh * MT + mt) * 16 + fr, p.M - 1) * p.lda + fq * 8;
#pragma unroll
            for (int i = 0; i < S2_MAXSTEPS; ++i) areg[i][mt] = *reinterpret_cast<const half8*>(ap + max(0, min(ks0 + i, ks1 - 1)) * 32);
        }
    }
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < S2_MAXSTEPS; ++i)
            if (ks0 + i < ks1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[i], areg[i][mt], acc[mt], 0, 0, 0);
            }
        if (tile + (int)gridDim.x < ntiles) load_w(tile + gridDim.x);
        float* pt = part_all[it & 1][rh];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(pt + ((wave * MT + mt) * 64 + lane) * 4) = acc[mt];
        // one barrier per tile: the buffers alternate, and a summing wave only reaches the barrier of tile it + 1
        // after it has finished reading tile it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave < MT) {
            const int mt = wave;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < S2_WAVES; ++w) t += *reinterpret_cast<const f32x4*>(pt + ((w * MT + mt) * 64 + lane) * 4);
            const int m = (rh * MT + mt) * 16 + fr;
            const int nb = tile * 16 + 4 * fq;
            if (m < p.M) {
                if (p.bias || p.gelu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (p.bias && nb + r < p.N) t[r] += (float)p.bias[nb + r];
                        if (p.gelu) t[r] = gelu_f(t[r]);
                    }
                }
                if (p.out_f && nb + 3 < p.N && (p.ldo & 3) == 0) {
                    *reinterpret_cast<f32x4*>(p.out_f + (long)m * p.ldo + nb) = t;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) {
                            if (p.out_f) p.out_f[(long)m * p.ldo + nb + r] = t[r];
                            else p.out_h[(long)m * p.ldo + nb + r] = (h16)t[r];
                        }
                }
            }
        }
    }
}

// The LayerNorm prologue of skinny2p_kernel<1, true> as a kernel of its own, for launches of more than 16 rows (the
// logits GEMV then takes the normalised rows from memory): the same thread-to-chunk assignment (16 threads per row, chunks
// t16, t16 + 16, ...), the same dot2 statistics, 16-lane DPP sums and fma_mix normalisation, rounded to fp16 once -- a
// row gets the BITS the fused prologue gives it, so its logits, token and log-probability do not depend on how many rows
// its pass holds (resln_kernel's two-pass fp32 LayerNorm differs in the last bit now and then).
__global__ __launch_bounds__(256) void ln_rows16_kernel(const h16* __restrict__ x, long ldx, const h16* __restrict__ g,
                                                        const h16* __restrict__ b, h16* __restrict__ y, long ldy, int M, int K) {
    const int tid = threadIdx.x, row = blockIdx.x * 16 + (tid >> 4), t16 = tid & 15, nch = K >> 3;
    const h16* xr = x + (long)min(row, M - 1) * ldx;
    half8 xv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) xv[i] = *reinterpret_cast<const half8*>(xr + min(t16 + 16 * i, nch - 1) * 8);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        ln_accum((t16 + 16 * i < nch) ? xv[i] : zero8, s1, s2);
    }
    s1 = sum16_dpp(s1);
    s2 = sum16_dpp(s2);
    const float mean = s1 / (float)K;
    const float rstd = rsqrtf(fmaxf(s2 / (float)K - mean * mean, 0.f) + 1e-5f);
    const float nmr = -mean * rstd;
    if (row >= M) return;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int c = t16 + 16 * i;
        if (c < nch) {
            const half8 g8 = *reinterpret_cast<const half8*>(g + c * 8);
            const half8 b8 = *reinterpret_cast<const half8*>(b + c * 8);
            *reinterpret_cast<half8*>(y + (long)row * ldy + c * 8) = ln_apply(xv[i], g8, b8, rstd, nmr);
        }
    }
}

// x_new = x + bias + sum_ky part[ky]   (or token + positional embedding at the start of a
// step), stored back in fp16; xn = LayerNorm(x_new).  One block per activation row.
__global__ __launch_bounds__(256) void resln_kernel(ResLnArgs p) {
    __shared__ float red[8];
    const int m = blockIdx.x, tid = threadIdx.x;
    constexpr int MAXE = 8;   // d <= 2048
    float v[MAXE];
    float s = 0.f;
    const h16* emb_row = nullptr;
    const h16* pos_row = nullptr;
    if (p.tokens) {
        const int pos = *p.d_pos;
        emb_row = p.emb + (long)p.tokens[(long)m * p.tok_ld + pos] * p.d;
        pos_row = p.pos + (long)pos * p.d;
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = tid + 256 * e;
        v[e] = 0.f;
        if (c < p.d) {
            float a;
            if (p.tokens) {
                a = (float)emb_row[c] + (float)pos_row[c];
            } else {
                a = (float)p.x[(long)m * p.d + c];
                if (p.bias) a += (float)p.bias[c];
                for (int k = 0; k < p.ksplit; ++k) a += p.part[((long)k * 16 + m) * p.ldp + c];
            }
            const h16 r = (h16)a;
            p.x[(long)m * p.d + c] = r;
            v[e] = (float)r;
            s += v[e];
        }
    }
    if (!p.xn) return;
    const float mean = block_sum(s, red) / (float)p.d;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = tid + 256 * e;
        if (c < p.d) {
            const float t = v[e] - mean;
            q += t * t;
        }
    }
    const float rstd = rsqrtf(block_sum(q, red) / (float)p.d + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = tid + 256 * e;
        if (c < p.d) p.xn[(long)m * p.d + c] = (h16)((v[e] - mean) * rstd * (float)p.g[c] + (float)p.b[c]);
    }
}

// row-major [N][K] -> [N / 16][K / 32][16][32] (SkinnyArgs::w_blocked), 8 elements per thread
template <typename T8>
__global__ __launch_bounds__(256) void pack_gemv_weight_kernel(const T8* __restrict__ w, T8* __restrict__ out, int N, int K) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;            // chunk (n, c): elements [8c, 8c + 8) of row n
    const int nch = K >> 3;
    if (i >= (long)N * nch) return;
    const int n = (int)(i / nch), c = (int)(i - (long)n * nch);
    const long o = (((long)(n >> 4) * (K >> 5) + (c >> 2)) * 16 + (n & 15)) * 4 + (c & 3);
    out[o] = w[i];
}

}  // namespace

static bool s2_walks(int M, int N, int ksplit) {
    return ksplit == 1 && M <= 64 && (N + 15) / 16 >= 2048;
}
bool skinny2_can_fuse_ln(int M, int N, int K) { return s2_walks(M, N, 1) && M <= 16 && K <= 1280 && (K & 7) == 0; }

hipError_t launch_skinny2(const Skinny2Args& a, hipStream_t s) {
    if (a.M > 64 && a.M <= 128 && a.ksplit == 1 && !a.ln_g) {
        // more than 64 rows (128-row passes): two launches of <= 64 rows -- the weights cross the fabric once per launch,
        // i.e. per 64 rows, as for a 64-row pass; every row keeps its bits
        Skinny2Args lo = a, hi = a;
        lo.M = 64;
        hi.M = a.M - 64;
        hi.A = a.A + 64 * a.lda;
        if (a.out_h) hi.out_h = a.out_h + 64 * a.ldo;
        if (a.out_f) hi.out_f = a.out_f + 64 * a.ldo;
        const hipError_t e = launch_skinny2(lo, s);
        return e != hipSuccess ? e : launch_skinny2(hi, s);
    }
    if ((a.K & 31) || a.M < 1 || a.M > 64 || a.ksplit < 1 || (a.M > 16 && a.ksplit > 1)) return hipErrorInvalidValue;
    if (((a.K >> 5) + a.ksplit * S2_WAVES - 1) / (a.ksplit * S2_WAVES) > S2_MAXSTEPS) return hipErrorInvalidValue;
    const int ntiles = (a.N + 15) / 16;
    if (s2_walks(a.M, a.N, a.ksplit)) {
        // several tiles per block, every block the same number (+-1), all blocks resident at once (132 VGPRs: 3 per CU)
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipErrorInvalidValue;
            n_cu = v;
        }
        const int per_cu = a.M <= 16 ? 3 : (a.M <= 32 ? 2 : 1);
        const int tpb = (ntiles + n_cu * per_cu - 1) / (n_cu * per_cu);
        const dim3 gridp((ntiles + tpb - 1) / tpb);
        if (a.ln_g) {
            if (a.M > 16 || a.K > 1280 || !a.ln_b) return hipErrorInvalidValue;
            hipLaunchKernelGGL((skinny2p_kernel<1, true>), gridp, dim3(256), 0, s, a, ntiles);
        } else if (a.M <= 16) {
            hipLaunchKernelGGL((skinny2p_kernel<1, false>), gridp, dim3(256), 0, s, a, ntiles);
        } else if (a.M <= 32) {
            hipLaunchKernelGGL((skinny2p_kernel<2, false>), gridp, dim3(256), 0, s, a, ntiles);
        } else {
            hipLaunchKernelGGL((skinny2p_kernel<2, false, 2>), gridp, dim3(512), 0, s, a, ntiles);
        }
        return hipGetLastError();
    }
    if (a.ln_g) return hipErrorInvalidValue;   // the fused LayerNorm exists in the tile-walking kernel only
    const dim3 grid(ntiles, a.ksplit);
    switch ((a.M + 15) / 16) {
        case 1: hipLaunchKernelGGL(skinny2_kernel<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(skinny2_kernel<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(skinny2_kernel<3>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(skinny2_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_ln_rows16(const h16* x, long ldx, const h16* g, const h16* b, h16* y, long ldy, int M, int K, hipStream_t s) {
    if (M < 1 || K > 1280 || (K & 7) || !g || !b) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_rows16_kernel, dim3((M + 15) / 16), dim3(256), 0, s, x, ldx, g, b, y, ldy, M, K);
    return hipGetLastError();
}

hipError_t launch_resln(const ResLnArgs& a, int M, hipStream_t s) {
    if (a.d > 2048) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resln_kernel, dim3(M), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_pack_gemv_weight(const void* w, void* out, int N, int K, int elem_bytes, hipStream_t s) {
    if (!w || !out || N < 16 || (N & 15) || K < 32 || (K & 31) || (elem_bytes != 1 && elem_bytes != 2)) return hipErrorInvalidValue;
    const long n = (long)N * (K >> 3);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (elem_bytes == 2)
        hipLaunchKernelGGL(pack_gemv_weight_kernel<half8>, grid, dim3(256), 0, s, (const half8*)w, (half8*)out, N, K);
    else
        hipLaunchKernelGGL(pack_gemv_weight_kernel<uint2>, grid, dim3(256), 0, s, (const uint2*)w, (uint2*)out, N, K);
    return hipGetLastError();
}
