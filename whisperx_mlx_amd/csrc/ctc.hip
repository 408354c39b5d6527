// CTC forced-alignment DP on the GPU, bit-exact float32.
//   get_trellis            /root/reference/whisperx/alignment.py:387-404
//   get_wildcard_emission  /root/reference/whisperx/alignment.py:407-437
//   backtrack_beam         /root/reference/whisperx/alignment.py:500-579
// One block per segment.  Everything the sequential parts touch per frame lives in LDS: the blank column, the
// wildcard (max non-blank) score and trellis column 0 for all T frames, the token ids, two trellis rows, and the
// emissions of the next 32 frames (double-buffered: the following chunk is fetched into registers while the
// current one is consumed), so a trellis step costs LDS reads + one barrier instead of an L2 round trip.  Every row
// is also written to HBM for the backtrack.  Column 0 is torch.cumsum of the blank log-probs, which on CPU
// accumulates in float64 and rounds each prefix to float32 -- reproduced here so the trellis is bit-identical.
// Backtrack: the reference copies whole Python lists per beam; here each step keeps (token, parent, prob) per beam
// slot and the winner is unwound through the parent links -- same candidate order (stay before change, beam
// order), same stable descending sort on the trellis-lookup score, same +-inf skips.
#include <mutex>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int CTC_MAXBEAM = 8;

constexpr int CTC_CHUNK = 32;          // frames of emissions staged per LDS buffer
constexpr int WIN_R = 16, WIN_C = 32;  // backtrack window (frames x columns) kept in LDS
constexpr int CTC_STAGE_REGS = 8;      // floats per thread of the chunk in flight: chunk_frames * V <= 256 * 8

__host__ __device__ inline int ctc_chunk_frames(int V) {
    const int f = (256 * CTC_STAGE_REGS) / V;
    return f < 1 ? 0 : (f > CTC_CHUNK ? CTC_CHUNK : f);   // 0: vocabulary too large to stage, read emissions from L2
}

// WB: compile-time bound of the beam width (2 or CTC_MAXBEAM) -- sizes the broadcasts of the beam step
// STAGED: the emissions of the next frames are staged in LDS (vocabulary small enough); the trellis step then holds no
// global load at all -- a load would make the compiler wait with vmcnt(0), i.e. also for the previous step's row stores
template <int WB, bool STAGED>
__global__ __launch_bounds__(256) void ctc_kernel(CtcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int sidx = blockIdx.x, tid = threadIdx.x;
    const int T = p.T[sidx], N = p.N[sidx];
    const int CH = STAGED ? ctc_chunk_frames(p.V) : 0;
    float* row0 = reinterpret_cast<float*>(smem);          // [Nmax]
    float* row1 = row0 + p.Nmax;                           // [Nmax]
    int* tok_l = reinterpret_cast<int*>(row1 + p.Nmax);    // [Nmax]
    float* blank_l = reinterpret_cast<float*>(tok_l + p.Nmax);   // [Tmax]
    float* wild_l = blank_l + p.Tmax;                      // [Tmax]
    float* col0_l = wild_l + p.Tmax;                       // [Tmax]
    float* embuf = col0_l + p.Tmax;                        // [2][CH * V]
    const float* __restrict__ em = p.logp + (long)sidx * p.seg_stride;
    const int* __restrict__ tok = p.tokens + (long)sidx * p.Nmax;
    float* __restrict__ tr = p.trellis + (long)sidx * p.Tmax * p.Nmax;
    float* __restrict__ wild = p.wild + (long)sidx * p.Tmax;
    int* ok = p.ok + sidx;
    if (T <= 0 || N <= 0) {
        if (tid == 0) *ok = 0;
        return;
    }
    // wildcard score per frame (max over non-blank labels), blank column, token ids
    for (int t = tid; t < T; t += blockDim.x) {
        float m = -INFINITY;
        for (int v = 0; v < p.V; ++v)
            if (v != p.blank) m = fmaxf(m, em[(long)t * p.V + v]);
        wild[t] = m;
        wild_l[t] = m;
        blank_l[t] = em[(long)t * p.V + p.blank];
    }
    for (int j = tid; j < N; j += blockDim.x) tok_l[j] = tok[j];
    // first chunk of emissions
    const int chunk_elems = CH * p.V;
    float stage[CTC_STAGE_REGS];
    if (STAGED) {
        const int n0 = min(chunk_elems, T * p.V);
        for (int i = tid; i < n0; i += blockDim.x) embuf[i] = em[i];
    }
    __syncthreads();
    // column 0 (float64 running sum rounded per prefix), then the +inf tail
    if (tid == 0) {
        double acc = 0.0;
        col0_l[0] = 0.f;
        for (int t = 1; t < T; ++t) {
            acc += (double)blank_l[t];
            col0_l[t] = (float)acc;
        }
        const int first_inf = (N > 1) ? max(T - N + 1, 0) : 0;
        for (int t = first_inf; t < T; ++t) col0_l[t] = INFINITY;
    }
    __syncthreads();
    for (int t = tid; t < T; t += blockDim.x) tr[(long)t * p.Nmax] = col0_l[t];
    // row 0
    for (int j = tid; j < N; j += blockDim.x) {
        const float v = (j == 0) ? col0_l[0] : -INFINITY;
        row0[j] = v;
        tr[j] = v;
    }
    __syncthreads();
    float* prev = row0;
    float* cur = row1;
    int tk_r[4];                                           // token ids of this thread's first four columns
#pragma unroll
    for (int q = 0; q < 4; ++q) tk_r[q] = (tid + q * 256 < N) ? tok_l[tid + q * 256] : 0;
    int tc = 0, chunk = 0;                                 // frame within the staged chunk, chunk index (t = chunk * CH + tc)
    for (int t = 0; t < T - 1; ++t) {
        const float* eb_row = embuf + (chunk & 1) * chunk_elems + tc * p.V;
        if (STAGED && tc == 0) {
            // fetch the following chunk into registers; it is parked in the other buffer at this chunk's last step
            const long base = (long)(chunk + 1) * chunk_elems;
            const long lim = (long)T * p.V;
#pragma unroll
            for (int r = 0; r < CTC_STAGE_REGS; ++r) {
                const long i = base + tid + r * 256;
                stage[r] = (tid + r * 256 < chunk_elems && i < lim) ? em[i] : 0.f;
            }
        }
        const float eb = blank_l[t];
        const float w = wild_l[t];
        auto column = [&](int j, int tk) {
            float v;
            if (j == 0) {
                v = col0_l[t + 1];
            } else {
                float e;
                if constexpr (STAGED) e = (tk < 0) ? w : eb_row[tk];
                else e = (tk < 0) ? w : em[(long)t * p.V + tk];
                v = fmaxf(prev[j] + eb, prev[j - 1] + e);
                tr[(long)(t + 1) * p.Nmax + j] = v;
            }
            cur[j] = v;
        };
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (tid + q * 256 < N) column(tid + q * 256, tk_r[q]);
        for (int j = tid + 1024; j < N; j += 256) column(j, tok_l[j]);
        if (STAGED && tc == CH - 1) {
            float* dst = embuf + ((chunk + 1) & 1) * chunk_elems;
#pragma unroll
            for (int r = 0; r < CTC_STAGE_REGS; ++r)
                if (tid + r * 256 < chunk_elems) dst[tid + r * 256] = stage[r];
            tc = 0;
            ++chunk;
        } else if (STAGED) {
            ++tc;
        }
        // LDS-only barrier: __syncthreads() would also wait for this step's trellis-row stores to reach L2
        // (a full memory round trip per frame); nothing reads them before the __threadfence() after the loop
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* tmp = prev; prev = cur; cur = tmp;
    }
    __threadfence();
    __syncthreads();
    if (tid >= 64) return;

    // ---- beam backtrack: wave 0 only.  Every lane carries the same beam state (uniform control flow); the lanes
    // share the work of refilling a window of the trellis and of the matching token emissions in LDS -- WIN_R frames
    // back from the current one, WIN_C columns left of the leading beam -- so that a step reads LDS instead of
    // waiting for dependent HBM/L2 loads.  A beam may move one column left per frame, so one refill lasts WIN_R
    // steps unless the beams are spread wider than the window (then that beam reads global memory directly).
    const int lane = tid;
    const int W = min(p.beam, WB);
    int* bp_tok = p.bp_tok + (long)sidx * (p.Tmax + 1) * CTC_MAXBEAM;
    int* bp_par = p.bp_par + (long)sidx * (p.Tmax + 1) * CTC_MAXBEAM;
    float* bp_prob = p.bp_prob + (long)sidx * (p.Tmax + 1) * CTC_MAXBEAM;
    float* win_tr = embuf + 2 * chunk_elems;               // [WIN_R][WIN_C]
    float* win_e = win_tr + WIN_R * WIN_C;                 // [WIN_R][WIN_C]
    int* bpl_tp = reinterpret_cast<int*>(win_e + WIN_R * WIN_C);        // [(Tmax + 1) * W]: token | parent << 28
    float* bpl_pr = reinterpret_cast<float*>(bpl_tp + (p.Tmax + 1) * W);   // [(Tmax + 1) * W]
    const bool bp_lds = p.bp_in_lds != 0;
    auto bp_store = [&](int st, int k, int tokv, int parv, float probv) {
        if (bp_lds) {
            bpl_tp[st * W + k] = tokv | ((parv & 7) << 28);
            bpl_pr[st * W + k] = probv;
        } else {
            bp_tok[st * CTC_MAXBEAM + k] = tokv;
            bp_par[st * CTC_MAXBEAM + k] = parv;
            bp_prob[st * CTC_MAXBEAM + k] = probv;
        }
    };
    // Lane c < 2 * nb is candidate c of the reference's list: beam c / 2, "stay" (even) or "change" (odd) -- the order
    // in which alignment.py:531-557 appends them.  The stable descending sort becomes a rank computed from
    // broadcasts of the 2 * W scores; nothing is indexed dynamically, so the beam state stays in registers.
    const int k = lane >> 1, kind = lane & 1;
    int nb = 1, step = 0, t = T - 1;
    int j = N - 1;                                          // column of this lane's beam (lanes < 2 * nb)
    j = (lane < 2 * CTC_MAXBEAM) ? j : 0;
    if (lane == 0) bp_store(0, 0, N - 1, 0, blank_l[T - 1]);   // records carry log-probabilities; exp() after the unwind
    int rt = -1, c0 = 0;                                   // window: frames rt .. rt-WIN_R+1, columns c0 .. c0+WIN_C-1
    bool dead = false;
    while (nb > 0 && __builtin_amdgcn_readfirstlane(j) > 0) {
        bool valid = false;
        float score = -INFINITY, prob_log = 0.f;
        int col = 0;
        if (t > 0) {
            const int row = t - 1;
            int minj = __builtin_amdgcn_readfirstlane(j), maxj = minj;
#pragma unroll
            for (int kk = 1; kk < WB; ++kk) {
                const int jk = __builtin_amdgcn_readlane(j, 2 * kk);
                minj = (kk < nb) ? min(minj, jk) : minj;
                maxj = (kk < nb) ? max(maxj, jk) : maxj;
            }
            const bool rows_ok = rt >= 0 && row <= rt && row > rt - WIN_R;
            const bool cols_ok = maxj <= c0 + WIN_C - 1 && (minj == 0 || minj - 1 >= c0);
            if (!rows_ok || (!cols_ok && maxj - minj <= WIN_C - 6)) {
                rt = row;
                c0 = maxj - WIN_C + 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier window reads are done
#pragma unroll
                for (int i = 0; i < WIN_R * WIN_C / 64; ++i) {
                    const int idx = lane + i * 64;
                    const int wr = rt - idx / WIN_C, wc = c0 + idx % WIN_C;
                    float tv = 0.f, ev = 0.f;
                    if (wr >= 0 && wc >= 0 && wc < N) {
                        tv = tr[(long)wr * p.Nmax + wc];
                        const int tk = tok_l[wc];
                        ev = (tk < 0) ? wild_l[wr] : em[(long)wr * p.V + tk];
                    }
                    win_tr[idx] = tv;
                    win_e[idx] = ev;
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if (lane < 2 * nb && (kind == 0 || j > 0)) {
                const int wbase = (rt - row) * WIN_C;
                col = j - kind;
                const int wc = col - c0, wcj = j - c0;
                score = (wc >= 0 && wc < WIN_C) ? win_tr[wbase + wc] : tr[(long)row * p.Nmax + col];
                if (kind == 0) {
                    prob_log = blank_l[row];
                } else if (wcj >= 0 && wcj < WIN_C) {
                    prob_log = win_e[wbase + wcj];
                } else {
                    const int tk = tok_l[j];
                    prob_log = (tk < 0) ? wild_l[row] : em[(long)row * p.V + tk];
                }
                valid = !isinf(score);
            }
        }
        const unsigned long long vmask = __ballot(valid);
        const int nc = __popcll(vmask);
        nb = __builtin_amdgcn_readfirstlane(min(nc, W));
        ++step;
        --t;
        if (nb == 0) { dead = true; break; }
        // rank in the stable descending order of the valid candidates
        int rank = 0;
#pragma unroll
        for (int c = 0; c < 2 * WB; ++c) {
            const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, score), c));
            rank += (((vmask >> c) & 1) && (sc > score || (sc == score && c < lane))) ? 1 : 0;
        }
        int src = 0;                                        // candidate that becomes this lane's beam k
#pragma unroll
        for (int kk = 0; kk < WB; ++kk) {
            const unsigned long long m = __ballot(valid && rank == kk);
            if (k == kk && m) src = (int)__ffsll((long long)m) - 1;
        }
        const int newj = __shfl(col, src);
        const float newp = __shfl(prob_log, src);
        if (lane < 2 * nb) {
            j = newj;
            if (kind == 0) bp_store(step, k, newj, src >> 1, newp);
        }
    }
    if (dead || nb == 0) {
        if (lane == 0) *ok = 0;
        return;
    }
    int* ptok = p.path_tok + (long)sidx * p.Tmax;
    float* pscore = p.path_score + (long)sidx * p.Tmax;
    // remaining frames stay on the final token with blank probability (:574-577)
    const int jfin = __builtin_amdgcn_readfirstlane(j);
    for (int tt = t - lane; tt > 0; tt -= 64) {
        ptok[tt - 1] = jfin;
        pscore[tt - 1] = expf(blank_l[tt - 1]);
    }
    // unwind beam 0 (serial parent chase, lane 0): step s <-> time index T-1-s; log-probabilities parked in LDS
    float* plog_l = col0_l;                                // column 0 is no longer needed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the other lanes' beam records have landed
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int kq = 0;
        for (int s2 = step; s2 >= 0; --s2) {
            const int tt = T - 1 - s2;
            if (bp_lds) {
                const int tp = bpl_tp[s2 * W + kq];
                ptok[tt] = tp & 0x0fffffff;
                plog_l[tt] = bpl_pr[s2 * W + kq];
                kq = (tp >> 28) & 7;
            } else {
                ptok[tt] = bp_tok[s2 * CTC_MAXBEAM + kq];
                plog_l[tt] = bp_prob[s2 * CTC_MAXBEAM + kq];
                kq = bp_par[s2 * CTC_MAXBEAM + kq];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int tt = T - 1 - step + lane; tt < T; tt += 64) pscore[tt] = expf(plog_l[tt]);
    if (lane != 0) return;
    *ok = 1;
}

// Fallback for segments whose per-frame arrays do not fit in LDS (minutes of audio in one segment): the same
// algorithm with only the two trellis rows in LDS and everything else read from global memory; the backtrack is a
// single thread.  Slow per frame, but unbounded in T.
__global__ __launch_bounds__(256) void ctc_plain_kernel(CtcArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int sidx = blockIdx.x, tid = threadIdx.x;
    const int T = p.T[sidx], N = p.N[sidx];
    float* row0 = reinterpret_cast<float*>(smem);          // [Nmax]
    float* row1 = row0 + p.Nmax;                           // [Nmax]
    const float* __restrict__ em = p.logp + (long)sidx * p.seg_stride;
    const int* __restrict__ tok = p.tokens + (long)sidx * p.Nmax;
    float* __restrict__ tr = p.trellis + (long)sidx * p.Tmax * p.Nmax;
    float* __restrict__ wild = p.wild + (long)sidx * p.Tmax;
    int* ok = p.ok + sidx;
    if (T <= 0 || N <= 0) {
        if (tid == 0) *ok = 0;
        return;
    }
    // wildcard score per frame: max over non-blank labels
    for (int t = tid; t < T; t += blockDim.x) {
        float m = -INFINITY;
        for (int v = 0; v < p.V; ++v)
            if (v != p.blank) m = fmaxf(m, em[(long)t * p.V + v]);
        wild[t] = m;
    }
    // column 0 (float64 running sum rounded per prefix), then the +inf tail
    if (tid == 0) {
        double acc = 0.0;
        tr[0] = 0.f;
        for (int t = 1; t < T; ++t) {
            acc += (double)em[(long)t * p.V + p.blank];
            tr[(long)t * p.Nmax] = (float)acc;
        }
        const int first_inf = (N > 1) ? max(T - N + 1, 0) : 0;
        for (int t = first_inf; t < T; ++t) tr[(long)t * p.Nmax] = INFINITY;
    }
    __threadfence_block();
    __syncthreads();
    // row 0
    for (int j = tid; j < N; j += blockDim.x) {
        const float v = (j == 0) ? tr[0] : -INFINITY;
        row0[j] = v;
        tr[j] = v;
    }
    __syncthreads();
    float* prev = row0;
    float* cur = row1;
    for (int t = 0; t < T - 1; ++t) {
        const float eb = em[(long)t * p.V + p.blank];
        const float w = wild[t];
        for (int j = tid; j < N; j += blockDim.x) {
            float v;
            if (j == 0) {
                v = tr[(long)(t + 1) * p.Nmax];
            } else {
                const int tk = tok[j];
                const float e = (tk < 0) ? w : em[(long)t * p.V + tk];
                v = fmaxf(prev[j] + eb, prev[j - 1] + e);
                tr[(long)(t + 1) * p.Nmax + j] = v;
            }
            cur[j] = v;
        }
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
    }
    __threadfence_block();
    __syncthreads();
    if (tid != 0) return;

    // ---- beam backtrack (single thread; T steps)
    const int W = min(p.beam, CTC_MAXBEAM);
    int* bp_tok = p.bp_tok + (long)sidx * (p.Tmax + 1) * CTC_MAXBEAM;
    int* bp_par = p.bp_par + (long)sidx * (p.Tmax + 1) * CTC_MAXBEAM;
    float* bp_prob = p.bp_prob + (long)sidx * (p.Tmax + 1) * CTC_MAXBEAM;
    int nb = 1, step = 0, t = T - 1;
    int bj[CTC_MAXBEAM];
    bj[0] = N - 1;
    bp_tok[0] = N - 1;
    bp_par[0] = -1;
    bp_prob[0] = expf(em[(long)(T - 1) * p.V + p.blank]);
    bool dead = false;
    while (nb > 0 && bj[0] > 0) {
        int cj[2 * CTC_MAXBEAM], cpar[2 * CTC_MAXBEAM];
        float cs[2 * CTC_MAXBEAM], cprob[2 * CTC_MAXBEAM];
        int nc = 0;
        if (t > 0) {
            const float p_stay = em[(long)(t - 1) * p.V + p.blank];
            for (int k = 0; k < nb; ++k) {
                const int j = bj[k];
                const int tk = tok[j];
                const float p_change = (tk < 0) ? wild[t - 1] : em[(long)(t - 1) * p.V + tk];
                const float stay = tr[(long)(t - 1) * p.Nmax + j];
                const float change = (j > 0) ? tr[(long)(t - 1) * p.Nmax + j - 1] : -INFINITY;
                if (!isinf(stay)) { cj[nc] = j; cpar[nc] = k; cs[nc] = stay; cprob[nc] = expf(p_stay); ++nc; }
                if (j > 0 && !isinf(change)) { cj[nc] = j - 1; cpar[nc] = k; cs[nc] = change; cprob[nc] = expf(p_change); ++nc; }
            }
        }
        // stable descending sort, keep W
        int order[2 * CTC_MAXBEAM];
        for (int a = 0; a < nc; ++a) {
            int pos = a;
            while (pos > 0 && cs[order[pos - 1]] < cs[a]) { order[pos] = order[pos - 1]; --pos; }
            order[pos] = a;
        }
        nb = min(nc, W);
        ++step;
        --t;
        if (nb == 0) { dead = true; break; }
        for (int k = 0; k < nb; ++k) {
            const int a = order[k];
            bj[k] = cj[a];
            bp_tok[step * CTC_MAXBEAM + k] = cj[a];
            bp_par[step * CTC_MAXBEAM + k] = cpar[a];
            bp_prob[step * CTC_MAXBEAM + k] = cprob[a];
        }
    }
    if (dead || nb == 0) { *ok = 0; return; }
    // unwind beam 0: step s <-> time index T-1-s
    int* ptok = p.path_tok + (long)sidx * p.Tmax;
    float* pscore = p.path_score + (long)sidx * p.Tmax;
    {
        int k = 0;
        for (int s2 = step; s2 >= 0; --s2) {
            const int tt = T - 1 - s2;
            ptok[tt] = bp_tok[s2 * CTC_MAXBEAM + k];
            pscore[tt] = bp_prob[s2 * CTC_MAXBEAM + k];
            k = bp_par[s2 * CTC_MAXBEAM + k];
        }
    }
    // remaining frames stay on the final token with blank probability (:574-577)
    const int jfin = bj[0];
    for (int tt = t; tt > 0; --tt) {
        ptok[tt - 1] = jfin;
        pscore[tt - 1] = expf(em[(long)(tt - 1) * p.V + p.blank]);
    }
    *ok = 1;
}

}  // namespace

hipError_t launch_ctc(const CtcArgs& a0, hipStream_t s) {
    CtcArgs a = a0;
    if (a.beam < 1 || a.beam > CTC_MAXBEAM) return hipErrorInvalidValue;
    const int W = a.beam;
    size_t lds = ((size_t)3 * a.Nmax + (size_t)3 * a.Tmax + (size_t)2 * ctc_chunk_frames(a.V) * a.V + 2 * WIN_R * WIN_C) * sizeof(float);
    const size_t bp = (size_t)(a.Tmax + 1) * W * 8;
    a.bp_in_lds = (lds + bp <= 150 * 1024 && a.Nmax < (1 << 28)) ? 1 : 0;   // else the beam records go through the HBM scratch
    if (a.bp_in_lds) lds += bp;
    if (lds > 150 * 1024) {
        const size_t lds_plain = (size_t)2 * a.Nmax * sizeof(float);
        if (lds_plain > 64 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(ctc_plain_kernel, dim3(a.S), dim3(256), lds_plain, s, a);
        return hipGetLastError();
    }
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        const void* kernels[4] = {reinterpret_cast<const void*>(ctc_kernel<2, true>), reinterpret_cast<const void*>(ctc_kernel<2, false>),
                                  reinterpret_cast<const void*>(ctc_kernel<CTC_MAXBEAM, true>),
                                  reinterpret_cast<const void*>(ctc_kernel<CTC_MAXBEAM, false>)};
        for (const void* kf : kernels)
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (attr_err != hipSuccess) return attr_err;
    const bool staged = ctc_chunk_frames(a.V) > 0;
    const dim3 grid(a.S), block(256);
    if (a.beam <= 2) {
        if (staged) hipLaunchKernelGGL((ctc_kernel<2, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((ctc_kernel<2, false>), grid, block, lds, s, a);
    } else {
        if (staged) hipLaunchKernelGGL((ctc_kernel<CTC_MAXBEAM, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((ctc_kernel<CTC_MAXBEAM, false>), grid, block, lds, s, a);
    }
    return hipGetLastError();
}
