// Attention kernels, d_head = 64, fp16 storage / fp32 softmax and accumulation.
//
// (1) attn_full_kernel -- non-causal self attention over T keys (Whisper encoder
//     T=1500, wav2vec2 T<=1500 with per-sequence lengths).  Flash style: one block
//     = 128 queries of one (batch, head); 4 waves x 32 queries; K/V tiles of 64
//     keys double-buffered in LDS.  Both products run on MFMA 32x32x16 f16 in the
//     "transposed" orientation so that a lane always owns one query:
//         S^T[key][q] = K[key][:] . Q[q][:]        (A = K tile,  B = Q^T regs)
//         O^T[d][q]  += V^T[d][key] * P^T[key][q]  (A = V^T tile, B = P^T = the S^T
//                                                   accumulators, no LDS round trip)
//     so the running max / sum / rescale are per-lane scalars.  V arrives
//     pre-transposed (VT[b][h*64+d][t], written by the V projection GEMM).
//
// (2) dec_attn_kernel -- single-query decode attention (decoder self attention over
//     the growing cache, cross attention over the 1500 encoder keys).  Pure HBM
//     streaming of K and V rows in their natural [t][d] layout: a wave instruction
//     fetches 8 keys x 128 B (8 lanes x 16 B per key; non-temporal for the cross K/V,
//     which are read once per position), dot products are v_dot2 + a DPP reduction
//     over the 8 lanes, scores go to LDS, softmax, then P.V streams V the same way.
//     Cross attention is key-split over blocks; the partials are merged in the same
//     launch through tagged 8-byte granules (see dec_cross_attn_kernel) and the
//     pre-softmax scores of the alignment heads are captured for the DTW word timing.
//     Self attention takes the step's own k, v straight from the QKV output and loads
//     every cached key in one trip.
#include "common.h"
#include "kernels.h"
#include "decode_dev.h"
#include <cstdlib>
#include <type_traits>

namespace {

// ------------------------------------------------------------------ (1) full attention
constexpr int KT = 64;             // keys per tile
constexpr int KSTR = 72;           // K tile row stride in halves (144 B: conflict-free b128 reads)
constexpr int VSTR = 68;           // V^T tile row stride in halves (136 B: conflict-free b64 reads)
constexpr int KTILE_B = KT * KSTR * 2;
constexpr int VTILE_B = 64 * VSTR * 2;

// PERSIST (AttnArgs::max_blocks): a capped grid, every block walks units blockIdx.x, + gridDim.x, ... (see gemm_8phase_kernel)
// Three blocks per CU (round 5): at launch_bounds(256, 2) the compiler takes 188 registers -- two waves per SIMD, with which
// the MFMA -> v_exp -> conversion -> MFMA chain of one wave is all the other has to hide behind.  At three it keeps 168 and
// spills 16 dwords to scratch; measured (profiles/r05_ab_attn_occupancy.txt): attention 1 081 -> 1 049 us at 64 rows, the
// encoder 156.1 -> 153.5 ms, the same bits.  Four blocks (128 registers, 296 bytes of scratch) is 5x slower.
#ifndef LAB_ATTN_OCC
#define LAB_ATTN_OCC 3
#endif
template <bool PERSIST>
__global__ __launch_bounds__(256, LAB_ATTN_OCC) void attn_full_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (KTILE_B + VTILE_B)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, XCD-aware: all query tiles of one (batch, head) run on the same XCD so that its
    // K / V^T panels (384 KiB at T = 1500) are fetched into that XCD's L2 once, not once per XCD
    const int nq = (p.T + 127) / 128;
    const int n_units = nq * p.H * p.B;
  for (int unit = blockIdx.x; unit < n_units; unit += PERSIST ? (int)gridDim.x : n_units) {
    const int logical = xcd_remap(unit, n_units);
    const int qt = logical % nq, bh = logical / nq;
    const int b = bh / p.H, h = bh - b * p.H;
    const int len = p.lens ? p.lens[b] : p.T;
    const int q0 = qt * 128 + wave * 32;
    if (qt * 128 >= len) continue;   // whole block beyond this sequence (uniform)

    const h16* __restrict__ Q = p.Q + (long)b * p.strideQ + h * 64;
    const h16* __restrict__ K = p.K + (long)b * p.strideK + h * 64;
    const h16* __restrict__ VT = p.VT + (long)b * p.strideVT + (long)h * 64 * p.ldvt;

    const int lr = lane & 31, lh = lane >> 5;
    // Q^T fragments (B operand): Q[q0+lr][16*s + 8*lh + j]
    half8 qf[4];
    {
        const int qi = min(q0 + lr, len - 1);
        const h16* qp = Q + (long)qi * p.ldq + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const half8*>(qp + 16 * s);
    }

    // staging: K tile 64 rows x 8 chunks(16 B) = 512 chunks; V^T tile the same
    int krow[2], kch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        krow[i] = c >> 3;
        kch[i] = c & 7;
    }
    half8 kreg[2], vreg[2];
    const unsigned ldk24 = (unsigned)p.ldk;   // < 2^24 (checked at launch), row index < 2^24
    auto gload = [&](int t0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // 24-bit multiply + 32-bit offset from the uniform base (full-rate VALU; the 64-bit row * ld product costs
            // four quarter-rate v_mul_lo_u32 per tile in a loop that is bound by VALU issue)
            const unsigned kr = (unsigned)min(t0 + krow[i], len - 1);
            kreg[i] = *reinterpret_cast<const half8*>(K + (__umul24(kr, ldk24) + (unsigned)(kch[i] * 8)));
            vreg[i] = *reinterpret_cast<const half8*>(VT + (long)krow[i] * p.ldvt + t0 + kch[i] * 8);
        }
    };
    auto sstore = [&](int buf) {
        char* kb = smem + buf * (KTILE_B + VTILE_B);
        char* vb = kb + KTILE_B;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<half8*>(kb + (krow[i] * KSTR + kch[i] * 8) * 2) = kreg[i];
            // V^T rows are only 8-byte aligned (136 B stride): two 8-byte stores
            half4 lo = {vreg[i][0], vreg[i][1], vreg[i][2], vreg[i][3]};
            half4 hi = {vreg[i][4], vreg[i][5], vreg[i][6], vreg[i][7]};
            char* dst = vb + (krow[i] * VSTR + kch[i] * 8) * 2;
            *reinterpret_cast<half4*>(dst) = lo;
            *reinterpret_cast<half4*>(dst + 8) = hi;
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    const float c2 = 0.125f * 1.44269504088896340736f;   // d_head^-0.5 * log2(e)
    // The scores leave the MFMAs ready for v_exp_f32 (round 5): the scale is IN the query -- the encoder's Q|K GEMM
    // multiplies its Q columns by c2 before its one rounding to fp16 (GemmArgs::xscale; callers that hand over raw queries
    // get them scaled here, a second rounding) -- and the running reference m_ref (log2 domain, an INTEGER, so every move
    // of it rescales by an exact power of two) enters through one more MFMA per subtile: A = a column of ones, B = -m_ref
    // as fp16 hi + lo in the lanes of its query.  That MFMA replaces 32 v_fma per tile on the vector port this loop is bound by.
    if (!p.q_prescaled) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (h16)((float)qf[s][j] * c2);
    }
    half8 onesf, mrf;
#pragma unroll
    for (int j = 0; j < 8; ++j) onesf[j] = mrf[j] = (h16)0.f;
    onesf[0] = onesf[1] = (h16)(lh == 0 ? 1.f : 0.f);
    float m_ref = 0.f, l_run = 0.f;   // reference of the exponentials (scaled scores, log2 domain; integer valued), running sum

    const int ntiles = (len + KT - 1) / KT;
    gload(0);
    sstore(0);
    // a use of the Q fragments BEFORE the loop: otherwise their loads count as possibly pending at the loop header and
    // the compiler's vmcnt(3..0) in front of the QK MFMAs also waits, every iteration, for the prefetch issued just above
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(qf[s]));
    __syncthreads();

    // one tile of 64 keys.  MASK (the sequence's last tile: keys >= len get -inf) is a compile-time property of the call:
    // left as a run-time test the compiler turned the masking into 60 compares + 60 selects on EVERY tile (round 5: the
    // kernel got slower by more than the per-tile maximum had cost), so the last tile is peeled off the loop instead
#ifdef WX_LAB_ENV
    // lab: core-clock stamps of units 0 / 1500 / 3000 (first wave), per key tile: 0 start | 1 scores in registers | 2 exponentials
    // done | 3 PV MFMAs issued | 4 next tile parked in LDS (vmcnt wait + ds_writes) | 5 barrier passed
    const int lab_sel = (unit == 0 ? 0 : unit == 1500 ? 1 : unit == 3000 ? 2 : -1);
    auto stamp = [&](int t, int k) {
        if (p.lab_stamps && lab_sel >= 0 && wave == 0 && lane == 0 && t < 24)
            p.lab_stamps[(lab_sel * 24 + t) * 8 + k] = __builtin_amdgcn_s_memtime();
    };
#define WX_STAMP(t, k) stamp(t, k)
#else
#define WX_STAMP(t, k)
#endif
    auto tile = [&](int t, auto MASKC) {
        constexpr bool MASK = decltype(MASKC)::value;
        WX_STAMP(t, 0);
        if (t + 1 < ntiles) gload((t + 1) * KT);
        const char* kb = smem + (t & 1) * (KTILE_B + VTILE_B);
        const char* vb = kb + KTILE_B;

        // S^T = K . Q^T - m_ref   (two 32-key subtiles; log2 domain, relative to the reference)
        f32x16 s0, s1;
        auto scores = [&]() {
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(onesf, mrf, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(onesf, mrf, s1, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const half8 ka = *reinterpret_cast<const half8*>(kb + ((lr)*KSTR + 16 * s + 8 * lh) * 2);
                const half8 kb2 = *reinterpret_cast<const half8*>(kb + ((32 + lr) * KSTR + 16 * s + 8 * lh) * 2);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kb2, qf[s], s1, 0, 0, 0);
            }
            // mask keys >= len (only the last tile can contain them)
            if (MASK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KT + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= len) s0[r] = -INFINITY;
                    if (key + 32 >= len) s1[r] = -INFINITY;
                }
            }
        };
        float psum;
        auto exps = [&]() {
            float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);       // raw v_exp_f32 (-inf -> 0; exp2f() adds ~6 range-handling ops per call)
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
                psum0 += s0[r];
                psum1 += s1[r];
            }
            psum = psum0 + psum1;
        };
        // online softmax for this lane's query (keys are split over the two half-waves).  This section, not the MFMAs, is
        // the critical path of the kernel (VALU issue slots), so the common tile is: 32 v_exp, 32 adds, 16 packed
        // conversions and ONE comparison.  No per-tile maximum: the exponentials are taken against the standing reference
        // and the reference moves only when a lane's sum of them says that some score has run far above it (sum > 2^14, so
        // every probability stays below 2^14: exact in fp32, same relative precision in the fp16 P operand, no overflow
        // there) -- then, and on the first tile, the scores are formed again, the tile's maximum taken, the reference
        // raised to its ceiling and the running sum and outputs scaled by that exact power of two.  Written as a loop
        // around ONE copy of the scores and the exponentials: two copies joined by a branch left the compiler ~50 register
        // moves on the common edge.
        // Plain f32 ops on purpose: this file is built with -fno-slp-vectorize because packed-f32 VALU
        // (v_pk_mul/add/fma_f32) issues several times slower than two scalar ops next to MFMAs on gfx950.
        bool adjust = (t == 0);
        for (;;) {
            scores();
            if (adjust) {
                float mloc0 = fmaxf(s0[0], s0[1]), mloc1 = fmaxf(s1[0], s1[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) {
                    mloc0 = fmaxf(fmaxf(mloc0, s0[r]), s0[r + 1]);
                    mloc1 = fmaxf(fmaxf(mloc1, s1[r]), s1[r + 1]);
                }
                float mloc = fmaxf(mloc0, mloc1);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));    // both half-waves of a query share its reference
                // first tile: the reference becomes the ceiling of the tile's maximum (whatever its sign); later: only ever raised
                const float u = (t == 0) ? ceilf(mloc) : fmaxf(ceilf(mloc), 0.f);
                m_ref += u;
                {
                    const h16 hi = (h16)(-m_ref);
                    const h16 lo = (h16)(-m_ref - (float)hi);
                    mrf[0] = lh == 0 ? hi : (h16)0.f;
                    mrf[1] = lh == 0 ? lo : (h16)0.f;
                }
                if (t > 0) {
                    const float alpha = __builtin_amdgcn_exp2f(-u);
                    l_run *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o0[r] *= alpha;
                        o1[r] *= alpha;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s0[r] -= u;
                    s1[r] -= u;
                }
            }
#ifdef WX_LAB_ENV
            asm volatile("" ::"v"(s0[15]), "v"(s1[15]));
            WX_STAMP(t, 1);
#endif
            exps();
            if (adjust || !__any(!(psum <= 16384.f))) break;     // wave-uniform; the comparison also catches inf / NaN
            adjust = true;
        }
        l_run += psum;
        WX_STAMP(t, 2);
        // O^T += V^T . P^T : k-step (kt, s2) covers keys kt*32 + 16*s2 + {8*(j>>2) + 4*lh + (j&3)}
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (h16)(kt == 0 ? s0[8 * s2 + j] : s1[8 * s2 + j]);
                const int koff = kt * 32 + 16 * s2 + 4 * lh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* vp = vb + ((db * 32 + lr) * VSTR + koff) * 2;
                    const half4 v_lo = *reinterpret_cast<const half4*>(vp);
                    const half4 v_hi = *reinterpret_cast<const half4*>(vp + 16);
                    const half8 vf = {v_lo[0], v_lo[1], v_lo[2], v_lo[3], v_hi[0], v_hi[1], v_hi[2], v_hi[3]};
                    if (db == 0)
                        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o0, 0, 0, 0);
                    else
                        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o1, 0, 0, 0);
                }
            }
        }
        // keep the parking of the prefetched tile HERE: left to itself the scheduler hoists these ds_writes (and with
        // them the vmcnt wait on the loads issued at the top of this iteration) up between the QK MFMAs
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        WX_STAMP(t, 3);
        if (t + 1 < ntiles) sstore((t + 1) & 1);
        WX_STAMP(t, 4);
        __syncthreads();
        WX_STAMP(t, 5);
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile(t, std::false_type{});
    tile(ntiles - 1, std::true_type{});

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + lr;
    if (qi < len) {
        h16* op = p.O + (long)b * p.strideO + (long)qi * p.ldo + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 a, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = (h16)(o0[4 * g + r] * inv);
                c[r] = (h16)(o1[4 * g + r] * inv);
            }
            *reinterpret_cast<half4*>(op + 8 * g + 4 * lh) = a;
            *reinterpret_cast<half4*>(op + 32 + 8 * g + 4 * lh) = c;
        }
    }
    if (PERSIST) __syncthreads();
  }
}
#undef WX_STAMP

// ------------------------------------------------------------------ (2) decode attention
constexpr int DEC_MAXKEYS = 1536;

struct DecAttnCore {
    const float* q_part; long q_ldp; int q_ksplit; const h16* q_bias;   // optional: query from split-K partials
    const h16* q;        // this (b,h): 64 halves
    const h16* K; long ldk;   // rows [t][64-slice]
    const h16* V; long ldv;
    int k_begin, k_end;  // key range of this block
    const h16* k_last = nullptr;   // optional: the row of key k_end-1 comes from here instead of K / V
    const h16* v_last = nullptr;   // (the step's own k, v straight from the QKV output, not via the cache)
};

// Shared body: scores -> LDS, softmax stats, P.V.  Returns (via refs) the block's
// unnormalised output (valid in threads 0..63: o[d]), max and sum.
__device__ __forceinline__ void dec_attn_body(const DecAttnCore& c, float* sc, float* red, float* ored,
                                              float& m_out, float& l_out, float& o_out, float* cap, int cap_ok) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const int ks = lane >> 3, dc = lane & 7;
    // The query is requested first and not touched before the first keys are requested too (a load behind a branch, or
    // one that is scaled right away, is waited for at once: one more serial round trip per block); the d_head^-0.5
    // scale, a power of two and therefore exact, goes onto the fp32 scores instead.
    half8 qh = *reinterpret_cast<const half8*>((c.q ? c.q : c.q_bias) + dc * 8);
    if (c.q_part) {
        // query = bias + sum of the producing GEMV's split-K partial tiles (fixed order), rounded
        // through fp16 like the stored activation would have been
        const h16* qb = c.q_bias + dc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = (float)qb[j];
            for (int k = 0; k < c.q_ksplit; ++k) a += c.q_part[(long)k * 16 * c.q_ldp + dc * 8 + j];
            qh[j] = (h16)a;
        }
    }
    const int nkeys = c.k_end - c.k_begin;
    const int niter = (nkeys + 8 * nwave - 1) / (8 * nwave);
    constexpr int U = 4;   // loads in flight per wave: the stream is latency bound with one
    // phase 1: scores
    for (int it = 0; it < niter; it += U) {
        half8 kh[U];
        int kl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kl[u] = ((it + u) * nwave + wave) * 8 + ks;
            kh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(c.K + (long)(c.k_begin + min(kl[u], nkeys - 1)) * c.ldk + dc * 8));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float acc = sum8_dpp(dot8_f16(qh, kh[u])) * 0.125f;
            if (dc == 0 && kl[u] < nkeys) {
                sc[kl[u]] = acc;
                if (cap_ok) cap[c.k_begin + kl[u]] = acc;
            }
        }
    }
    // the first trip of values does not depend on the scores: requested here, it flies during the softmax reductions
    half8 vpre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int klp = (u * nwave + wave) * 8 + ks;
        vpre[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(c.V + (long)(c.k_begin + min(klp, nkeys - 1)) * c.ldv + dc * 8));
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int i = tid; i < nkeys; i += blockDim.x) mx = fmaxf(mx, sc[i]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int i = tid; i < nkeys; i += blockDim.x) {
        const float e = __expf(sc[i] - mx);
        sc[i] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    __syncthreads();
    // phase 2: o[d] = sum_key p[key] V[key][d]
    float ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int klp = (u * nwave + wave) * 8 + ks;
        const float pk = (klp < nkeys) ? sc[klp] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = fmaf(pk, (float)vpre[u][j], ov[j]);
    }
    for (int it = U; it < niter; it += U) {
        half8 vh[U];
        int kl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kl[u] = ((it + u) * nwave + wave) * 8 + ks;
            vh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(c.V + (long)(c.k_begin + min(kl[u], nkeys - 1)) * c.ldv + dc * 8));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pk = (kl[u] < nkeys) ? sc[kl[u]] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = fmaf(pk, (float)vh[u][j], ov[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ov[j] += __shfl_xor(ov[j], 8, 64);
        ov[j] += __shfl_xor(ov[j], 16, 64);
        ov[j] += __shfl_xor(ov[j], 32, 64);
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ored[wave * 64 + lane * 8 + j] = ov[j];
    }
    __syncthreads();
    float o = 0.f;
    if (tid < 64)
        for (int w = 0; w < nwave; ++w) o += ored[w * 64 + tid];
    m_out = mx;
    l_out = sum;
    o_out = o;
}

// Single-pass variant: online softmax per 8-lane key slot, K and V of U key groups in flight
// together, no block barrier until the final cross-wave merge (the two-pass body above idles
// HBM while the block sits in its softmax barriers).  Returns the same (max, sum, o) triple.
template <int U>
__device__ __forceinline__ void dec_attn_online(const DecAttnCore& c, float* wred, float& m_out, float& l_out,
                                                float& o_out, float* cap, int cap_ok) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6;
    const int ks = lane >> 3, dc = lane & 7;
    // The query is requested first and NOT touched until the first keys have been requested too: a load behind a branch,
    // or one that is scaled right away, is waited for at once -- one more serial memory round trip in a kernel that is
    // made of nothing else.  The d_head^-0.5 scale (a power of two: exact) is applied to the fp32 scores instead.
    half8 qh = *reinterpret_cast<const half8*>((c.q ? c.q : c.q_bias) + dc * 8);
    if (c.q_part) {
        const h16* qb = c.q_bias + dc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = (float)qb[j];
            for (int k = 0; k < c.q_ksplit; ++k) a += c.q_part[(long)k * 16 * c.q_ldp + dc * 8 + j];
            qh[j] = (h16)a;
        }
    }
    const int nkeys = c.k_end - c.k_begin;
    const int niter = (nkeys + 8 * nwave - 1) / (8 * nwave);
    float m = -INFINITY, l = 0.f, ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = 0.f;
    for (int it = 0; it < niter; it += U) {
        half8 kh[U], vh[U];
        int kl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kl[u] = ((it + u) * nwave + wave) * 8 + ks;
            const long row = c.k_begin + min(kl[u], nkeys - 1);
            const h16* kp = (c.k_last && kl[u] >= nkeys - 1) ? c.k_last : c.K + row * c.ldk;
            kh[u] = *reinterpret_cast<const half8*>(kp + dc * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long row = c.k_begin + min(kl[u], nkeys - 1);
            const h16* vp = (c.v_last && kl[u] >= nkeys - 1) ? c.v_last : c.V + row * c.ldv;
            vh[u] = *reinterpret_cast<const half8*>(vp + dc * 8);
        }
        float sc[U];
        float m_new = m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float acc = sum8_dpp(dot8_f16(qh, kh[u])) * 0.125f;
            const bool ok = kl[u] < nkeys;
            if (ok && dc == 0 && cap_ok) cap[c.k_begin + kl[u]] = acc;
            sc[u] = ok ? acc : -INFINITY;
            m_new = fmaxf(m_new, sc[u]);
        }
        if (m_new > -INFINITY) {
            const float alpha = __expf(m - m_new);      // m = -inf -> 0
            float psum = 0.f, pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pk = __expf(sc[u] - m_new);
                psum += pk;
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = fmaf(pk, (float)vh[u][j], pv[j]);
            }
            l = l * alpha + psum;
#pragma unroll
            for (int j = 0; j < 8; ++j) ov[j] = ov[j] * alpha + pv[j];
            m = m_new;
        }
    }
    // merge the 8 key slots of this wave
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
        const float M = fmaxf(m, mo);
        const float a = (m > -INFINITY) ? __expf(m - M) : 0.f, b = (mo > -INFINITY) ? __expf(mo - M) : 0.f;
        l = l * a + lo * b;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = ov[j] * a + __shfl_xor(ov[j], off, 64) * b;
        m = M;
    }
    // merge the waves: wred[wave][66] = {m, l, o[64]}
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) wred[wave * 66 + 2 + lane * 8 + j] = ov[j];
        if (lane == 0) {
            wred[wave * 66] = m;
            wred[wave * 66 + 1] = l;
        }
    }
    __syncthreads();
    float M = -INFINITY;
    for (int w = 0; w < nwave; ++w) M = fmaxf(M, wred[w * 66]);
    float L = 0.f, o = 0.f;
    for (int w = 0; w < nwave; ++w) {
        const float mw = wred[w * 66];
        const float a = (mw > -INFINITY) ? __expf(mw - M) : 0.f;
        L += a * wred[w * 66 + 1];
        if (tid < 64) o += a * wred[w * 66 + 2 + tid];
    }
    m_out = M;
    l_out = L;
    o_out = o;
}

__global__ __launch_bounds__(256) void dec_self_attn_kernel(DecSelfAttnArgs p, const h16* __restrict__ knew,
                                                            const h16* __restrict__ vnew, long ldnew) {
#pragma clang fp contract(off)      // as in skinny.hip / declayer.hip: this arithmetic has a twin (xattn_role) that must give the same bits
    __shared__ float ored[4 * 66];
#ifdef LAB_GEMV_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    // a finished row: nothing it computes is looked at again (uniform per block).  Its flag is loaded UNCONDITIONALLY (from
    // an address that is always valid) next to the position: two scalar loads, one wait.  Behind a branch on the pointer
    // the compiler chains pointer -> flag -> position into three dependent round trips (+0.5 us per launch).
    const int* flag_p = p.done ? p.done + b : p.d_pos;
    const int pos = *p.d_pos;
    const int flag = *flag_p;
    if (p.done && flag && pos >= 0) return;      // (pos >= 0 always: it keeps the position's load in front of the branch, next to the flag's)
    h16* kc = p.kc + (long)b * p.cache_stride + h * 64;
    h16* vc = p.vc + (long)b * p.cache_stride + h * 64;
    // append this step's k,v slice for (b,h) to the cache for the later steps; this step's attention takes the row
    // straight from the QKV output.  The 16 lanes request their piece here, with everything else, and store it at the very
    // end: as "load; store" up front the two branches cost wave 0 two serial memory round trips before its first key.
    const h16* kn = knew + (long)b * ldnew + h * 64;
    const h16* vn = vnew + (long)b * ldnew + h * 64;
    const bool appender = tid < 16;
    const h16* app_src = tid < 8 ? kn + tid * 8 : vn + (tid & 7) * 8;
    h16* app_dst = (tid < 8 ? kc : vc) + (long)pos * p.d + (tid & 7) * 8;
    const half8 app = *reinterpret_cast<const half8*>(app_src);   // every lane (a load behind a branch is waited for at once)
    #ifdef LAB_NO_SELFKV     // lab builds only (tools/build_lab.py): the self-attention reads one cached key instead of pos + 1
    DecAttnCore c{nullptr, 0, 0, nullptr, p.q + (long)b * p.ldq + h * 64, kc, (long)p.d, vc, (long)p.d, 0, 1};
#else
    DecAttnCore c{nullptr, 0, 0, nullptr, p.q + (long)b * p.ldq + h * 64, kc, (long)p.d, vc, (long)p.d, 0, min(pos + 1, 512)};
#endif
    c.k_last = kn;
    c.v_last = vn;
    float m, l, o;
    // 4 waves x 8 keys x U keys per trip: one trip (one memory round trip) covers 64 / 128 keys.  (U = 8 -- 256 keys a
    // trip -- is 104 registers against 63: four blocks per CU instead of eight, and the 2 240 blocks of a 112-row launch
    // then take three rounds instead of one and a bit; beyond 128 keys the second trip costs less than that.)
    if (pos + 1 > 64)
        dec_attn_online<4>(c, ored, m, l, o, nullptr, 0);
    else
        dec_attn_online<2>(c, ored, m, l, o, nullptr, 0);
    if (tid < 64) p.out[act_index(b, h * 64 + tid, p.ldo, p.out_blocked)] = (h16)(o / l);
    __builtin_amdgcn_sched_barrier(0);
    if (appender) *reinterpret_cast<half8*>(app_dst) = app;
}

__global__ __launch_bounds__(512) void dec_cross_attn_kernel(DecCrossAttnArgs p, int nsplit, float* __restrict__ part) {
#pragma clang fp contract(off)      // as in skinny.hip / declayer.hip: this arithmetic has a twin (xattn_role) that must give the same bits
    __shared__ float sc[DEC_MAXKEYS];
    __shared__ float red[8];
    __shared__ float ored[8 * 66];
    const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z, tid = threadIdx.x;
    {   // a finished row: all of its split blocks return, nobody waits for anybody (flag loaded unconditionally, see above)
        const int* flag_p = p.done ? p.done + b : p.d_pos;
        const int flag = *flag_p;
        if (p.done && flag) return;
    }
    const int per = (((p.T + nsplit - 1) / nsplit) + 7) & ~7;
    const int k0 = sp * per, k1 = min(p.T, k0 + per);
    int cap_ok = 0;
    float* cap = nullptr;
    if (p.qk_out) {
        const int slot = p.cap_slot[h];
        const int row = *p.d_row;
        if (slot >= 0 && row >= 0 && row < p.cap_rows) {
            cap_ok = 1;
            cap = p.qk_out + (((long)b * p.n_cap + slot) * p.cap_rows + row) * p.T;
        }
    }
    DecAttnCore c{p.q_part ? p.q_part + (long)b * p.q_ldp + h * 64 : nullptr, p.q_ldp, p.q_ksplit,
                  p.q_bias ? p.q_bias + h * 64 : nullptr,
                  p.q ? p.q + (long)b * p.ldq + h * 64 : nullptr, p.K + (long)b * p.strideK + h * p.hstride, p.ldk,
                  p.V + (long)b * p.strideV + h * p.hstride, p.ldv, k0, k1};
    float m, l, o;
    // two-pass body: measured faster than the online variant for the 376-key cross-attention slices
    if (p.online)
        dec_attn_online<4>(c, ored, m, l, o, cap, cap_ok);
    else
        dec_attn_body(c, sc, red, ored, m, l, o, cap, cap_ok);
    if (nsplit == 1) {
        if (tid < 64) p.out[act_index(b, h * 64 + tid, p.ldo, p.out_blocked)] = (h16)(o / l);
    } else if (p.gran) {
        // Merge of the key splits by data-tagged granules: every partial word travels as ONE naturally
        // aligned 8-byte {value, tag} write-through (sc1) store, so a reader that sees the tag of this
        // (decode call, position, layer) has the value too -- no fence, no counter, and the producers
        // leave without waiting for anything.  The block of the LAST split merges: it is dispatched
        // after its producers (blockIdx.z is the slowest grid dimension), polls their granules with
        // sc1 loads (bounded: a give-up raises *d_err and poisons the row) and sums in split order, so
        // the result is independent of arrival order.  Replaces partial stores + vmcnt drain + ticket
        // atomic + read-back (4.4 us of dependent round trips at the end of every launch).
        const unsigned tag = (*p.d_epoch << 16) | 0x8000u | ((unsigned)(*p.d_pos) << 6) | (unsigned)p.layer;
        unsigned long long* g = p.gran + (((long)b * p.H + h) * nsplit + sp) * 66;
        auto pack = [&](float v) { return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v); };
        if (sp != nsplit - 1) {
            if (tid < 64) __hip_atomic_store(g + 2 + tid, pack(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                __hip_atomic_store(g, pack(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g + 1, pack(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (tid < 64) {
            const unsigned long long* base = p.gran + ((long)b * p.H + h) * nsplit * 66;
            float ms[4], ls[4], os[4];
            bool ok = true;
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                ms[s2] = -INFINITY; ls[s2] = 0.f; os[s2] = 0.f;
                if (s2 >= nsplit - 1) continue;
                unsigned long long gm = 0, gl = 0, go = 0;
                int spin = 0;
                for (; spin < (1 << 18); ++spin) {
                    gm = __hip_atomic_load(base + s2 * 66, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gl = __hip_atomic_load(base + s2 * 66 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    go = __hip_atomic_load(base + s2 * 66 + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(gm >> 32) == tag && (unsigned)(gl >> 32) == tag && (unsigned)(go >> 32) == tag) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                ok = ok && spin < (1 << 18);
                ms[s2] = __uint_as_float((unsigned)gm);
                ls[s2] = __uint_as_float((unsigned)gl);
                os[s2] = __uint_as_float((unsigned)go);
            }
            ms[3] = m; ls[3] = l; os[3] = o;          // own part is the last split: summed last
            float M = -INFINITY;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) M = fmaxf(M, ms[s2]);
            float L = 0.f, O = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const float w = (ms[s2] > -INFINITY) ? __expf(ms[s2] - M) : 0.f;   // absent splits contribute exactly 0
                L += w * ls[s2];
                O += w * os[s2];
            }
            if (!ok) {
                if (p.d_err) *p.d_err = 1;
                O = __builtin_nanf("");
            }
            p.out[act_index(b, h * 64 + tid, p.ldo, p.out_blocked)] = (h16)(O / L);
        }
    } else if (!p.tickets) {
        float* pp = part + (((long)b * p.H + h) * nsplit + sp) * 66;
        if (tid < 64) pp[2 + tid] = o;
        if (tid == 0) {
            pp[0] = m;
            pp[1] = l;
        }
    } else {
        // In-launch merge of the key-split partials (saves the separate combine launch).  Hand-off
        // form (MI355X per-XCD L2s are not coherent): every partial word is a write-through (sc1)
        // agent-scope store, every storing wave drains vmcnt, one lane then draws a ticket with a
        // relaxed agent-scope fetch_add; the block whose add returns nsplit-1 is last, and reads all
        // partials back with sc1 (agent-scope) loads only -- no plain load ever touches them.  The
        // partials are summed in split order, so the result does not depend on arrival order.
        unsigned* pp = reinterpret_cast<unsigned*>(part + (((long)b * p.H + h) * nsplit + sp) * 66);
        if (tid < 64) __hip_atomic_store(pp + 2 + tid, __float_as_uint(o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(pp, __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pp + 1, __float_as_uint(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int s_last;
        if (tid == 0) {
            unsigned* cnt = p.tickets + (long)b * p.H + h;
            const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == (unsigned)(nsplit - 1));
            if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        if (s_last && tid < 64) {
            unsigned* base = reinterpret_cast<unsigned*>(part + ((long)b * p.H + h) * nsplit * 66);
            float M = -INFINITY;
            for (int s2 = 0; s2 < nsplit; ++s2)
                M = fmaxf(M, __uint_as_float(__hip_atomic_load(base + s2 * 66, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
            float L = 0.f, O = 0.f;
            for (int s2 = 0; s2 < nsplit; ++s2) {
                const float ms = __uint_as_float(__hip_atomic_load(base + s2 * 66, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const float ls = __uint_as_float(__hip_atomic_load(base + s2 * 66 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const float os = __uint_as_float(__hip_atomic_load(base + s2 * 66 + 2 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const float w = __expf(ms - M);
                L += w * ls;
                O += w * os;
            }
            p.out[act_index(b, h * 64 + tid, p.ldo, p.out_blocked)] = (h16)(O / L);
        }
    }
}

__global__ void dec_attn_combine_kernel(const float* __restrict__ part, int nsplit, h16* __restrict__ out, long ldo, int H,
                                        int out_blocked) {
#pragma clang fp contract(off)      // as in skinny.hip / declayer.hip: this arithmetic has a twin (xattn_role) that must give the same bits
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;   // 64 threads
    const float* pp = part + ((long)b * H + h) * nsplit * 66;
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pp[s * 66]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float w = __expf(pp[s * 66] - m);
        l += w * pp[s * 66 + 1];
        o += w * pp[s * 66 + 2 + d];
    }
    out[act_index(b, h * 64 + d, ldo, out_blocked)] = (h16)(o / l);
}

}  // namespace

#ifdef WX_LAB_ENV
static constexpr size_t LAB_ATTN_STAMP_WORDS = 3 * 24 * 8;
static unsigned long long* g_lab_attn_stamps = nullptr;
extern "C" int wx_lab_read_attn_stamps(unsigned long long* out) {
    if (!g_lab_attn_stamps) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpy(out, g_lab_attn_stamps, LAB_ATTN_STAMP_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

hipError_t launch_attention(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
#ifdef WX_LAB_ENV
    static const int lab_stamps_on = WX_LAB_GETENV_INT("WX_ATTN_STAMPS", 0);
    if (lab_stamps_on) {
        if (!g_lab_attn_stamps) (void)hipMalloc(reinterpret_cast<void**>(&g_lab_attn_stamps), LAB_ATTN_STAMP_WORDS * sizeof(unsigned long long));
        if (g_lab_attn_stamps) (void)hipMemsetAsync(g_lab_attn_stamps, 0, LAB_ATTN_STAMP_WORDS * sizeof(unsigned long long), s);
        a.lab_stamps = g_lab_attn_stamps;
    }
#endif
    if (a.ldk >= (1 << 24) || a.T >= (1 << 24) || (long)a.T * a.ldk >= (1L << 31)) return hipErrorInvalidValue;
    const int units = ((a.T + 127) / 128) * a.H * a.B;
    if (a.max_blocks > 0 && a.max_blocks < units)
        hipLaunchKernelGGL(attn_full_kernel<true>, dim3(a.max_blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(attn_full_kernel<false>, dim3(units), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_dec_self_attn(const DecSelfAttnArgs& a, const h16* knew, const h16* vnew, long ldnew, hipStream_t s) {
    hipLaunchKernelGGL(dec_self_attn_kernel, dim3(a.H, a.B), dim3(256), 0, s, a, knew, vnew, ldnew);
    return hipGetLastError();
}

hipError_t launch_dec_cross_attn(const DecCrossAttnArgs& a, int nsplit, float* part, hipStream_t s, int threads) {
    if (a.T > DEC_MAXKEYS || (threads != 128 && threads != 256 && threads != 512)) return hipErrorInvalidValue;
    DecCrossAttnArgs a2 = a;
    a2.online = 0;       // the two-pass body: measured faster than the single-pass (online softmax) one for 750-key slices
    if (nsplit > 4 || nsplit < 2) a2.gran = nullptr;
    hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(a.H, a.B, nsplit), dim3(threads), 0, s, a2, nsplit, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || nsplit == 1 || a.tickets || a2.gran) return e;
    hipLaunchKernelGGL(dec_attn_combine_kernel, dim3(a.H, a.B), dim3(64), 0, s, part, nsplit, a.out, a.ldo, a.H,
                       a.out_blocked);
    return hipGetLastError();
}
