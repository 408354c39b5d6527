// CTC forced-alignment DP on the GPU, bit-exact float32.
//   get_trellis            /root/reference/whisperx/alignment.py:387-404
//   get_wildcard_emission  /root/reference/whisperx/alignment.py:407-437
//   backtrack_beam         /root/reference/whisperx/alignment.py:500-579
// One block per segment.  Trellis: lanes over the N tokens, sequential over the T
// frames (two LDS rows), every row also written to HBM for the backtrack.  Column 0
// is torch.cumsum of the blank log-probs, which on CPU accumulates in float64 and
// rounds each prefix to float32 -- reproduced here so the trellis is bit-identical.
// Backtrack: the reference copies whole Python lists per beam; here each step keeps
// (token, parent, prob) per beam slot and the winner is unwound through the parent
// links -- same candidate order (stay before change, beam order), same stable
// descending sort on the trellis-lookup score, same +-inf skips.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int CTC_MAXBEAM = 8;

__global__ __launch_bounds__(256) void ctc_kernel(CtcArgs p) {
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

hipError_t launch_ctc(const CtcArgs& a, hipStream_t s) {
    if (a.beam < 1 || a.beam > CTC_MAXBEAM) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * a.Nmax * sizeof(float);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ctc_kernel, dim3(a.S), dim3(256), lds, s, a);
    return hipGetLastError();
}
