// Fused launches of the decode layer: dependent stages of one decoder layer share a launch and hand their results
// over through data-tagged 8-byte granules instead of a kernel boundary, so that the HBM stream of the consuming stage
// (the cross-attention's keys: 61 MB per layer at 16 rows) is already in flight while the producing GEMV still runs.
//
//   dec_cq_xattn_kernel   [LayerNorm + cross-Q GEMV]  ->  [cross attention over the 1500 encoder keys]
//       blocks [0, G)        the GEMV of skinny_kernel<LN, 5, 8> (same code, same summation order); its epilogue
//                            publishes the query as granules {2 x fp16, tag} (write-through sc1 stores)
//       blocks [G, G + H*B)  one (batch row, head) each, 8 waves: waves 0-3 take the first half of the keys, waves 4-7
//                            the second half -- the two key splits of dec_cross_attn_kernel side by side in one block,
//                            merged through LDS in split order (bit-identical to the split launch + granule merge).
//                            Every wave requests ALL its keys (24 x 16 B per lane, non-temporal) and the first trip of
//                            values before it needs the query; wave 0 then polls the 32 granules of its (row, head).
//
// Forward progress does not depend on the producers: a consumer polls for its 32 granules for a short while (~100 us;
// the producers of its own launch have the lower block ids, are dispatched first and normally publish within ~6 us),
// and when they have not arrived -- consumers of one launch can hold the wave slots the producers of ANOTHER stream's
// launch need, while that launch's consumers hold theirs -- the block computes its query itself: it runs the GEMV role
// for the 64 columns of its head and the 16-row group of its row (same code, same k-split order: the same bits) and
// takes its row of the result through LDS.  Nothing is poisoned, nothing is decoded again; `n_selfq` counts the blocks
// that did so (wx_decode_stats).
// References: the reference's decoder step is one opaque mlx call (mlx_whisper_batch_decoder.py:70-72,84-86); the
// stage arithmetic follows skinny.hip / attention.hip.
// No implicit fused multiply-adds in this file: `a * b + c` stays a multiply and an add unless the source says fmaf().
// The same expression is compiled several times here (template instances, roles inlined into different kernels), and
// under the default contraction the compiler fused it in one copy and not in another -- with int8 weights the fused
// launch's GEMV role, its self-computed-query path and skinny_kernel then disagreed in the last bit now and then, and a
// row's log-probability depended on which copy had produced its query.  Everything hot already spells its FMAs out.
// (The last of those differences was not a contraction at all: fma + conversion to fp16 folded into one v_fma_mix*_f16 in
// one copy -- decode_dev.h, scale_bias.)
#pragma clang fp contract(off)
#include <cstdlib>
#include "common.h"
#include "decode_dev.h"
#include "kernels.h"

#ifdef LAB_NO_W            // lab builds only (tools/build_lab.py)
#define WX_LAB_W(x) 0
#else
#define WX_LAB_W(x) (x)
#endif

namespace {

// polls (~1 us each: one sc1 load to L2 and back + s_sleep) before the consumer computes the query itself
#ifdef DL_POLL_OVERRIDE          // lab builds only (tools/): a poll window long enough that the path is never taken
constexpr int DL_POLL = DL_POLL_OVERRIDE;
#else
constexpr int DL_POLL = 128;
#endif

__device__ __forceinline__ unsigned long long pack_h2(float a, float b, unsigned tag) {
    const half2v v = {(h16)a, (h16)b};
    return ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- role: LayerNorm + GEMV over 16 rows, 8 waves split K, query published as granules ------------------------------
// (skinny_kernel<true, 5, 8, Q8> with the store replaced; keep the two in step)
// qs_lds != null (the consumer's own computation of its query): the pairs of row `sel_row` of the group go to
// qs_lds[(column - col0) / 2] in LDS instead of the granule buffer
template <bool Q8>
__device__ __forceinline__ void gemv_ln_publish_role(SkinnyArgs p, int bx, int by, unsigned long long* __restrict__ gq,
                                                     unsigned tag, char* smem, unsigned* qs_lds = nullptr, int sel_row = -1,
                                                     int col0 = 0) {
    constexpr int STEPS = 5, WAVES = 8, MAXC = 5;
    float* part = reinterpret_cast<float*>(smem);                  // [8][64][4] f32 = 8 KiB
    h16* a_lds = reinterpret_cast<h16*>(smem + WAVES * 64 * 16);  // [16][K+8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tn = p.tile_n > 0 ? p.tile_n : 16;
    const int n0 = bx * tn;
    const int lda_s = p.K + 8;
    const int row0 = by * 16;
    p.A += (long)row0 * p.lda;
    p.M = min(16, p.M - row0);

    half8 xv[MAXC];
    {
        const h16* xr0 = p.A + (long)min(tid >> 5, p.M - 1) * p.lda;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) xv[c] = *reinterpret_cast<const half8*>(xr0 + min((tid & 31) + 32 * c, (p.K >> 3) - 1) * 8);
    }
    half8 gbv;
    {
        const int nch0 = p.K >> 3;
        const int t2 = min(tid, 2 * nch0 - 1);
        gbv = *reinterpret_cast<const half8*>((t2 < nch0 ? p.ln_g : p.ln_b - (long)nch0 * 8) + (long)t2 * 8);
    }
    const int nks = p.K >> 5;
    const int ks0 = (wave * nks) / WAVES, ks1 = ((wave + 1) * nks) / WAVES;
    const int nstep = ks1 - ks0;
    const int nrow = min(n0 + min(fr, tn - 1), p.N - 1);
    const long wrow = p.w_blocked ? ((long)(nrow >> 4) * (p.K >> 5) * 16 + (nrow & 15)) * 32 : (long)nrow * p.ldw;   // skinny.hip
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
    const int em = fr, enb = n0 + 4 * fq;
    half4 eb4 = {0, 0, 0, 0};
    const int ncol = max(0, min(4, min(tn - 4 * fq, p.N - enb)));
    if (wave == 0) {
        const int nc = min(enb, (p.N - 4) & ~3);
        eb4 = *reinterpret_cast<const half4*>(p.bias ? p.bias + nc : p.A);
    }
    f32x4 es4 = {1.f, 1.f, 1.f, 1.f};
    if (Q8 && wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) es4[r] = p.wscale[min(enb + r, p.N - 1)];
    }
    {
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
        h16* gb_lds = reinterpret_cast<h16*>(smem);
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
            const half8 af = *reinterpret_cast<const half8*>(a_lds + fr * lda_s + (ks0 + i) * 32 + fq * 8);
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
        if (em < p.M && ncol == 4) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = scale_bias(t[r], es4[r], p.bias ? (float)eb4[r] : 0.f);
#if defined(LAB_DUMP_Q8) && !defined(LAB_DUMP_Q8_NOSTORES)
            if (p.lab_dump && enb == 608 && !qs_lds) {
                float* dd = p.lab_dump + (p.lab_slot * 16 + row0 + em) * 8;
                dd[0] = t[0]; dd[1] = es4[0]; dd[2] = (float)eb4[0]; dd[3] = v[0]; dd[4] = t[1]; dd[5] = es4[1]; dd[6] = (float)eb4[1]; dd[7] = v[1];
            }
#endif
            // the fp16 values the unfused kernel would have stored, two per granule
            if (qs_lds) {
                if (em == sel_row) {
                    qs_lds[(enb - col0) >> 1] = (unsigned)pack_h2(v[0], v[1], 0u);
                    qs_lds[((enb - col0) >> 1) + 1] = (unsigned)pack_h2(v[2], v[3], 0u);
                }
            } else {
                unsigned long long* g = gq + (long)(row0 + em) * (p.N >> 1) + (enb >> 1);
                __hip_atomic_store(g, pack_h2(v[0], v[1], tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g + 1, pack_h2(v[2], v[3], tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- role: cross attention of one (row, head); both key splits in one block -----------------------------------------
constexpr int XA_SC = 768;      // scores per split (T <= 1536)
constexpr int XA_MLO = 68;      // max, sum, o[64] of the first split (+2: the query behind it stays 16-byte aligned for ds_read_b128)
template <int NKI, bool Q8>
__device__ __forceinline__ void xattn_role(const DecCrossAttnArgs& p, const SkinnyArgs& pg, size_t selfq_off, int* n_selfq,
                                           const unsigned long long* __restrict__ gq, int qn2,
                                           unsigned tag, int bh, char* smem, int polls = DL_POLL) {
    // whether this row has already emitted EOT: the (scalar) load goes out first and is looked at only after the first
    // key trips have been requested -- a load that is consumed right away is one more serial round trip per block
    const int* dflag = p.done ? p.done + bh / p.H : p.d_pos;      // always a valid address: the load is unconditional
    const int row_done = *dflag;
    float* sc = reinterpret_cast<float*>(smem);          // [2][768]
    float* red = sc + 2 * XA_SC;                         // [8]
    float* ored = red + 8;                               // [8][64]
    float* mlo = ored + 8 * 64;                          // [XA_MLO]: max, sum, o[64] of the first split
    unsigned* qs = reinterpret_cast<unsigned*>(mlo + XA_MLO);   // [32]: the query, 2 x fp16 per word (16-byte aligned)
    unsigned* qself = reinterpret_cast<unsigned*>(smem + selfq_off);   // [33]: the self-computed query + flag, beyond the GEMV role's LDS
    // Everything the role derives from the thread index lives in one struct that can be derived AGAIN: after the cold path
    // below (the block computing its query itself) it is, from a laundered thread index, so that none of it has to stay
    // alive -- i.e. be spilled -- across the GEMV in there.
    struct Ix {
        int tid, lane, wave, grp, wg, gtid, ks, dc, k0, nkeys, cap_ok;
        const h16* K;
        const h16* V;
        float* cap;
    };
    const int h = bh % p.H, b = bh / p.H;
    auto derive = [&](int t) {
        Ix x;
        x.tid = t; x.lane = t & 63; x.wave = t >> 6; x.grp = x.wave >> 2; x.wg = x.wave & 3; x.gtid = t & 255;
        x.ks = x.lane >> 3; x.dc = x.lane & 7;
        const int per = (((p.T + 1) / 2) + 7) & ~7;
        x.k0 = x.grp * per;
        x.nkeys = min(p.T, x.k0 + per) - x.k0;
        x.K = p.K + (long)b * p.strideK + h * p.hstride + (long)x.k0 * p.ldk;
        x.V = p.V + (long)b * p.strideV + h * p.hstride + (long)x.k0 * p.ldv;
        x.cap_ok = 0;
        x.cap = nullptr;
        if (p.qk_out) {
            const int slot = p.cap_slot[h];
            const int row = *p.d_row;
            if (slot >= 0 && row >= 0 && row < p.cap_rows) {
                x.cap_ok = 1;
                x.cap = p.qk_out + (((long)b * p.n_cap + slot) * p.cap_rows + row) * p.T + x.k0;
            }
        }
        return x;
    };
    Ix ix = derive(threadIdx.x);
#define tid ix.tid
#define lane ix.lane
#define wave ix.wave
#define grp ix.grp
#define wg ix.wg
#define gtid ix.gtid
#define ks ix.ks
#define dc ix.dc
#define k0 ix.k0
#define nkeys ix.nkeys
#define K ix.K
#define V ix.V
#define cap_ok ix.cap_ok
#define cap ix.cap
    // (0) the first PRE trips of this wave's keys and the first trip of values: in flight before the query exists
    // (all 24 trips would cost 96 VGPRs and a wave of residency; half of them already cover the GEMV role's run time)
    constexpr int PRE = NKI / 2, PRE0 = 4;
    half8 kreg[PRE];
#pragma unroll
    for (int it = 0; it < PRE0; ++it) {
        const int kl = (it * 4 + wg) * 8 + ks;
        kreg[it] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(K + (long)min(kl, nkeys - 1) * p.ldk + dc * 8));
    }
    // a finished row (EOT emitted): the rest of its 2 x 61 MB per layer is not streamed; its output is never looked at
    if (p.done && row_done) return;
#pragma unroll
    for (int it = PRE0; it < PRE; ++it) {
        const int kl = (it * 4 + wg) * 8 + ks;
        kreg[it] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(K + (long)min(kl, nkeys - 1) * p.ldk + dc * 8));
    }
    constexpr int U = 4;
    half8 vpre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int klp = (u * 4 + wg) * 8 + ks;
        vpre[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(V + (long)min(klp, nkeys - 1) * p.ldv + dc * 8));
    }
    // (1) the query of this (row, head): 32 granules, polled by wave 0 only (gq == null: the query is in memory already,
    // written by a GEMV launch in front of this one -- step variant 5)
    if (gq == nullptr) {
        if (wave == 0) {
            if (lane < 32) qs[lane] = reinterpret_cast<const unsigned*>(p.q + (long)b * p.ldq + h * 64)[lane];
            if (lane == 0) qself[32] = 0u;
        }
    } else if (wave == 0) {
        const unsigned long long* g = gq + (long)b * qn2 + h * 32 + (lane & 31);
        unsigned long long v = 0;
        bool ok = false;
        for (int spin = 0; spin < polls; ++spin) {
            v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(v >> 32) == tag;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
        }
        const bool all_ok = __all(ok);              // over the whole wave: outside any lane-dependent branch
        if (lane < 32) qs[lane] = (unsigned)v;
        if (lane == 0) qself[32] = all_ok ? 0u : 1u;
    }
    lds_barrier();
#ifdef DL_NO_FALLBACK             // lab builds only: the round-2 code generation for comparison (an expired poll is ignored)
    if (false) {
#else
    if (__builtin_amdgcn_readfirstlane((int)qself[32])) {
#endif
        // the producers have not delivered (see the file header): this block computes the 64 query columns of its head
        // for its row group with the GEMV role's own code -- block-uniform branch, all 8 waves take part -- and reads
        // its row back through LDS.  The keys requested above are requested again afterwards (cold path: their
        // registers are free for the GEMV meanwhile).
        __syncthreads();
        const int tn = pg.tile_n > 0 ? pg.tile_n : 16;
        for (int j = 0; j < 64 / tn; ++j) {
            gemv_ln_publish_role<Q8>(pg, h * (64 / tn) + j, b >> 4, nullptr, tag, smem, qself, b & 15, h * 64);
            __syncthreads();
        }
        if (tid < 32) qs[tid] = qself[tid];
        if (tid == 0 && n_selfq) atomicAdd(n_selfq, 1);
        // Everything derived from the thread index is derived again, from a laundered copy (see `Ix` above): kept alive
        // across the GEMV it would be spilled -- on the HOT path too, and a kernel with a scratch frame runs ~10 % slower
        // whether or not this path is ever taken (16 rows alone: 1 234x against 1 360x).  Then the keys are requested again.
        int t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        ix = derive(t2);
#pragma unroll
        for (int it = 0; it < PRE; ++it) {
            const int kl = (it * 4 + wg) * 8 + ks;
            kreg[it] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(K + (long)min(kl, nkeys - 1) * p.ldk + dc * 8));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int klp = (u * 4 + wg) * 8 + ks;
            vpre[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(V + (long)min(klp, nkeys - 1) * p.ldv + dc * 8));
        }
        __syncthreads();
    }
    const half8 qh = *reinterpret_cast<const half8*>(qs + dc * 4);
    // (2) scores (d_head^-0.5 = 0.125 on the fp32 score: exact): the remaining trips are requested first, six at a time
    constexpr int KU = 6;
    half8 kmore[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int kl = ((PRE + u) * 4 + wg) * 8 + ks;
        kmore[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(K + (long)min(kl, nkeys - 1) * p.ldk + dc * 8));
    }
#pragma unroll
    for (int it = 0; it < PRE; ++it) {
        const int kl = (it * 4 + wg) * 8 + ks;
        const float acc = sum8_dpp(dot8_f16(qh, kreg[it])) * 0.125f;
        if (dc == 0 && kl < nkeys) {
            sc[grp * XA_SC + kl] = acc;
            if (cap_ok) cap[kl] = acc;
        }
    }
#pragma unroll
    for (int base = PRE; base < NKI; base += KU) {
        half8 knext[KU];
        if (base + KU < NKI) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int kl = ((base + KU + u) * 4 + wg) * 8 + ks;
                knext[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(K + (long)min(kl, nkeys - 1) * p.ldk + dc * 8));
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int kl = ((base + u) * 4 + wg) * 8 + ks;
            const float acc = sum8_dpp(dot8_f16(qh, kmore[u])) * 0.125f;
            if (dc == 0 && kl < nkeys) {
                sc[grp * XA_SC + kl] = acc;
                if (cap_ok) cap[kl] = acc;
            }
        }
        if (base + KU < NKI) {
#pragma unroll
            for (int u = 0; u < KU; ++u) kmore[u] = knext[u];
        }
    }
    __syncthreads();
    // (3) softmax statistics of each split over its 4 waves (block_max / block_sum of the split launch)
    float* scg = sc + grp * XA_SC;
    float mx = -INFINITY;
    for (int i = gtid; i < nkeys; i += 256) mx = fmaxf(mx, scg[i]);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[grp * 4];
    for (int i = 1; i < 4; ++i) mx = fmaxf(mx, red[grp * 4 + i]);
    float sum = 0.f;
    for (int i = gtid; i < nkeys; i += 256) {
        const float e = __expf(scg[i] - mx);
        scg[i] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    {
        float r = 0.f;
        for (int i = 0; i < 4; ++i) r += red[grp * 4 + i];
        sum = r;
    }
    __syncthreads();
    // (4) o[d] = sum_key p[key] V[key][d]
    const int niter = (nkeys + 31) / 32;
    float ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int klp = (u * 4 + wg) * 8 + ks;
        const float pk = (klp < nkeys) ? scg[klp] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = fmaf(pk, (float)vpre[u][j], ov[j]);
    }
    for (int it = U; it < niter; it += U) {
        half8 vh[U];
        int kl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kl[u] = ((it + u) * 4 + wg) * 8 + ks;
            vh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(V + (long)min(kl[u], nkeys - 1) * p.ldv + dc * 8));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pk = (kl[u] < nkeys) ? scg[kl[u]] : 0.f;
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
    if (gtid < 64)
        for (int w = 0; w < 4; ++w) o += ored[(grp * 4 + w) * 64 + gtid];
    // (5) merge of the two splits, in split order (the arithmetic of the granule merge in dec_cross_attn_kernel)
    if (grp == 0 && gtid < 64) {
        mlo[2 + gtid] = o;
        if (gtid == 0) {
            mlo[0] = mx;
            mlo[1] = sum;
        }
    }
    __syncthreads();
    if (grp == 1 && gtid < 64) {
        const float ms[2] = {mlo[0], mx}, ls[2] = {mlo[1], sum}, os[2] = {mlo[2 + gtid], o};
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) M = fmaxf(M, ms[s2]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float w = (ms[s2] > -INFINITY) ? __expf(ms[s2] - M) : 0.f;
            L += w * ls[s2];
            O += w * os[s2];
        }
        p.out[act_index(b, h * 64 + gtid, p.ldo, p.out_blocked)] = (h16)(O / L);
    }
}

#undef tid
#undef lane
#undef wave
#undef grp
#undef wg
#undef gtid
#undef ks
#undef dc
#undef k0
#undef nkeys
#undef K
#undef V
#undef cap_ok
#undef cap

struct CqXattnArgs {
    SkinnyArgs g;
    DecCrossAttnArgs a;
    unsigned long long* gq;              // the GEMV role publishes here ...
    const unsigned long long* gq_poll;   // ... and the attention role polls here (the same buffer, except in the test hook)
    int g_tiles, n_groups;
    int selfq_off;                       // byte offset of the self-computed query in the dynamic LDS (behind the GEMV role's area)
    int* n_selfq;                        // counter: attention blocks that computed their query themselves
    unsigned long long* prof_slot;       // optional launch timer: where block 0 notes the launch's start (null: off)
    int polls;                           // polls (~1 us each) before an attention block computes its query itself (DL_POLL)
};

template <bool Q8>
__global__ __launch_bounds__(512, 4) void dec_cq_xattn_kernel(CqXattnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned tag = (*p.a.d_epoch << 16) | 0x8000u | ((unsigned)(*p.a.d_pos) << 6) | (unsigned)p.a.layer;
    const int nG = p.g_tiles * p.n_groups;
    const int bid = blockIdx.x;
    // launch timer (wx_tuning.profile_launches): block 0 -- the first to be dispatched -- notes when the launch
    // started, on the constant 100 MHz clock; the first block of the NEXT launch on the stream (the output projection,
    // skinny_kernel) notes when it starts in turn, adds the difference to the record and clears the note.  One store here,
    // one load and two stores there: atomics from every block (or wave) to one address serialise and tripled the launch.
    if (p.prof_slot && bid == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.prof_slot, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bid < nG)
        gemv_ln_publish_role<Q8>(p.g, bid % p.g_tiles, bid / p.g_tiles, p.gq, tag, smem);
    else
        xattn_role<24, Q8>(p.a, p.g, (size_t)p.selfq_off, p.n_selfq, p.gq_poll, p.g.N >> 1, tag, bid - nG, smem, p.polls);
}

}  // namespace

bool dec_cq_xattn_supported(const SkinnyArgs& g, const DecCrossAttnArgs& a) {
    const int tn = g.tile_n > 0 ? g.tile_n : 16;
    return g.ln_g && g.ln_b && (g.K & 31) == 0 && g.K <= 1280 && (tn == 8 || tn == 16) && g.N % tn == 0 && g.N == a.H * 64 &&
           a.T <= 1536 && a.T >= 64 && g.M >= 1 && g.M <= 128 && a.gran && a.d_pos && a.d_epoch;
}

hipError_t launch_dec_cq_xattn(const SkinnyArgs& g, const DecCrossAttnArgs& a, unsigned long long* gq, hipStream_t s,
                               const unsigned long long* gq_poll, int* n_selfq, bool q_in_memory,
                               unsigned long long* prof_slot) {
    if (!dec_cq_xattn_supported(g, a) || !gq) return hipErrorInvalidValue;
    static const int lab_polls = WX_LAB_GETENV_INT("WX_DL_POLL", -1);      // lab builds: the window swept (tools/r05_sweeps.sh, profiles/r05_ab_dl_poll.txt)
    CqXattnArgs p{g, a, gq, gq_poll ? gq_poll : gq, 0, 0, 0, n_selfq, prof_slot, lab_polls >= 0 ? lab_polls : DL_POLL};
    const int tn = g.tile_n > 0 ? g.tile_n : 16;
    p.g_tiles = g.N / tn;
    p.n_groups = (g.M + 15) / 16;
    if (q_in_memory) {            // only the attention role: the query was written to a.q by a GEMV launch in front
        if (!a.q) return hipErrorInvalidValue;
        p.g_tiles = 0;
        p.gq_poll = nullptr;
    }
    const size_t lds_g = 8 * 64 * 16 + (size_t)16 * (g.K + 8) * 2;
    const size_t lds_a = (2 * XA_SC + 8 + 8 * 64 + XA_MLO + 32) * sizeof(float);
    p.selfq_off = (int)((lds_g > lds_a ? lds_g : lds_a) + 15) & ~15;
    size_t lds = (size_t)p.selfq_off + 36 * sizeof(unsigned);
#ifdef WX_LAB_ENV
    // lab: one block per CU instead of two (does the launch leave the other passes' GEMV blocks room, and what does it cost alone?)
    static const int lab_lds_kb = WX_LAB_GETENV_INT("WX_XATTN_LDS_KB", 0);
    if (lab_lds_kb * 1024 > (int)lds) {
        lds = (size_t)lab_lds_kb * 1024;
        static bool attr_set = false;
        if (!attr_set) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(dec_cq_xattn_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void*>(dec_cq_xattn_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
    }
#endif
    const dim3 grid(p.g_tiles * p.n_groups + a.H * a.B);
    if (g.Wq)
        hipLaunchKernelGGL(dec_cq_xattn_kernel<true>, grid, dim3(512), lds, s, p);
    else
        hipLaunchKernelGGL(dec_cq_xattn_kernel<false>, grid, dim3(512), lds, s, p);
    return hipGetLastError();
}
