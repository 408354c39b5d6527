// Greedy sampling step fused with the logit filters.
//   BatchGreedyDecoder.update   /root/reference/mlx_whisper_batch_decoder.py:267-303
//   timestamp-probability rule  /root/reference/mlx_ultra_optimized_batch.py:38-71
//   SuppressBlank / SuppressTokens / ApplyTimestampRules: published Whisper rules
//   (third-party mlx-whisper at the reference boundary).
// One block per sequence, two streaming passes over the (B, n_vocab) fp32 logits
// (L2 resident): pass 1 masked max/argmax of the text and timestamp ranges, pass 2
// the two exp-sums.  Ties resolve to the lowest token id.  Everything the next step
// needs (the new token, sum_logprob, no_speech_prob) stays on the device: the
// decode loop never syncs with the host (contrast mlx_whisper_batch_decoder.py:56-57).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

enum {
    RULE_SUPPRESS_BLANK = 1, RULE_SUPPRESS_TOKENS = 2, RULE_TS_NOTIMESTAMPS = 4, RULE_TS_PAIRS = 8,
    RULE_TS_MONOTONE = 16, RULE_TS_INITIAL = 32, RULE_TS_PROB = 64
};

struct RowState {
    int n, first, last_ts, pen_ts, ts_bound, forced;
};

__device__ __forceinline__ bool suppressed_dyn(const SampleArgs& p, const RowState& r, int v) {
    if (r.forced && v == p.eot) return true;
    if (r.first) {
        if ((p.rules & RULE_SUPPRESS_BLANK) && (v == p.blank0 || v == p.blank1 || v == p.eot)) return true;
        if (p.rules & RULE_TS_INITIAL) {
            if (v < p.timestamp_begin) return true;
            if (p.max_initial_ts >= 0 && v > p.timestamp_begin + p.max_initial_ts) return true;
        }
    }
    if ((p.rules & RULE_TS_PAIRS) && r.last_ts) {
        if (r.pen_ts) {
            if (v >= p.timestamp_begin) return true;
        } else {
            if (v < p.eot) return true;
        }
    }
    if ((p.rules & RULE_TS_MONOTONE) && v >= p.timestamp_begin && v < r.ts_bound) return true;
    return false;
}

// The dynamic rules of suppressed_dyn() as two allowed id ranges (text [t_lo, t_hi), timestamps [s_lo, s_hi)) plus up to
// three singly banned ids: evaluated once per row, so the two sweeps over the vocabulary test a range instead of
// re-deriving every rule for every token (the sweeps were VALU bound: ~25 instructions per logit).
struct RowRanges {
    int t_lo, t_hi, s_lo, s_hi, ban0, ban1, ban2;
};

__device__ __forceinline__ RowRanges row_ranges(const SampleArgs& p, const RowState& r) {
    RowRanges g;
    g.t_lo = 0; g.t_hi = p.timestamp_begin; g.s_lo = p.timestamp_begin; g.s_hi = p.n_vocab;
    g.ban0 = g.ban1 = g.ban2 = -1;
    if (r.forced) g.ban0 = p.eot;
    if (r.first) {
        if (p.rules & RULE_SUPPRESS_BLANK) { g.ban0 = p.eot; g.ban1 = p.blank0; g.ban2 = p.blank1; }
        if (p.rules & RULE_TS_INITIAL) {
            g.t_hi = 0;
            if (p.max_initial_ts >= 0) g.s_hi = min(g.s_hi, p.timestamp_begin + p.max_initial_ts + 1);
        }
    }
    if ((p.rules & RULE_TS_PAIRS) && r.last_ts) {
        if (r.pen_ts) g.s_hi = g.s_lo;            // no timestamp may follow two timestamps
        else g.t_lo = max(g.t_lo, p.eot);         // a lone timestamp must be followed by a timestamp or EOT
    }
    if (p.rules & RULE_TS_MONOTONE) g.s_lo = max(g.s_lo, r.ts_bound);
    return g;
}

__device__ __forceinline__ bool allowed_dyn(const RowRanges& g, int v) {
    return ((v >= g.t_lo && v < g.t_hi) || (v >= g.s_lo && v < g.s_hi)) && v != g.ban0 && v != g.ban1 && v != g.ban2;
}

__device__ __forceinline__ bool suppressed(const SampleArgs& p, const RowState& r, int v) {
    return p.suppress[v] || suppressed_dyn(p, r, v);
}

// exp of a non-positive argument on the raw v_exp_f32 (one multiply + one transcendental; libm's expf is ~15 VALU
// instructions of range handling that an argument <= 0 never needs; -inf and underflow give 0)
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// S = 1: one block per row (any shape).  S > 1 (vectorisable rows only): S blocks per row, each with a contiguous
// S-th of the vocabulary in registers (RV vec4 per thread); a block's maxima, first positions and exp-sums (relative
// to its own maximum) go to `p.part`, and the last block of a row to finish (ticket) merges the S records in
// vocabulary order -- same first-position tie rule -- and carries on with the decision and the tail.  One CU cannot
// pull a 207 KB row and evaluate its 52 K masks faster than ~10 us; four can.
template <int S, int RV>
__global__ __launch_bounds__(1024) void sample_kernel(SampleArgs p) {
    __shared__ float sv[2][16];
    __shared__ int si[2][16];
    __shared__ float ssum[2][16];
    __shared__ int s_lastts_idx;
    __shared__ RowState rs;
    const int b = blockIdx.x, kblk = (S > 1) ? blockIdx.y : 0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* tok = p.tokens + (long)b * p.tok_ld;
    const float* __restrict__ lg = p.logits + (long)b * p.ldl;
    // The row (<= 13 x 1024 vec4 = 53248 logits) stays in registers across both passes: ONE sweep over memory, every
    // load of a thread in flight at once (the kernel runs on B CUs only: it is a chain of memory round trips), and the
    // sweep is requested before anything else: the position / token-history reads below ride in its shadow.
    const int nvec = p.n_vocab >> 2;
    const f32x4* __restrict__ lg4 = reinterpret_cast<const f32x4*>(lg);
    const uchar4* __restrict__ sup4 = reinterpret_cast<const uchar4*>(p.suppress);
    const bool vec_ok = ((p.ldl & 3) == 0) && ((reinterpret_cast<size_t>(p.suppress) & 3) == 0);
    // this block's share of the vec4 groups: [q_lo, q_hi)
    const int q_per = (nvec + S - 1) / S;
    const int q_lo = kblk * q_per, q_hi = min(q_lo + q_per, nvec);
    const bool in_regs = (S > 1) || (vec_ok && nvec <= RV * 1024 && blockDim.x == 1024);   // S > 1: checked at launch
    f32x4 xs[RV];
    uchar4 m4[RV];
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int q = min(q_lo + tid + 1024 * i, q_hi - 1);
            xs[i] = lg4[q];
            m4[i] = sup4[q];
        }
    }
    const int pos = *p.d_pos;
    const int n_active = p.n_active ? *p.n_active : p.B;   // loaded with the position: no wait of its own later
    const int n = pos + 1;                      // tokens so far (incl. prompt)
    __shared__ int s_next;
    // fused tail (see SampleArgs): embedding of the token at position n for the next step, then the position counters
    auto tail = [&](int next) {
        if (!p.emb) return;
        const h16* e = p.emb + (long)next * p.d;
        const h16* pe = p.decpos + (long)n * p.d;
        for (int c = tid; c < (p.d >> 3); c += blockDim.x) {
            const half8 a = *reinterpret_cast<const half8*>(e + c * 8);
            const half8 q = *reinterpret_cast<const half8*>(pe + c * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (h16)((float)a[j] + (float)q[j]);
            *reinterpret_cast<half8*>(p.x + (long)b * p.d + c * 8) = o;
        }
        if (tid == 0) {
            // every block read *d_pos in its first instructions; the last one to get here moves it
            __threadfence();
            if (atomicAdd(p.ticket, 1u) == gridDim.x - 1) {   // gridDim.x = rows: one deciding block per row
                *p.ticket = 0;
                *p.d_pos_w = pos + 1;
                *p.d_row = pos + 1 - (p.sample_begin - 1);
            }
        }
    };
    if (n < p.sample_begin) {                   // still feeding the prompt
        if (kblk == 0) tail(tok[n]);
        return;
    }

    // ---- per-row history state
    if (tid == 0) s_lastts_idx = -1;
    __syncthreads();
    {
        int best = -1;
        for (int i = p.sample_begin + tid; i < n; i += blockDim.x)
            if (tok[i] >= p.timestamp_begin) best = i;
        if (best >= 0) atomicMax(&s_lastts_idx, best);
    }
    __syncthreads();
    if (tid == 0) {
        const int len = n - p.sample_begin;
        RowState r;
        r.n = n;
        r.first = (len == 0);
        r.last_ts = (len >= 1) && tok[n - 1] >= p.timestamp_begin;
        r.pen_ts = (len < 2) || tok[n - 2] >= p.timestamp_begin;
        r.ts_bound = 0;
        if (s_lastts_idx >= 0) {
            const int last = tok[s_lastts_idx];
            r.ts_bound = (r.last_ts && !r.pen_ts) ? last : last + 1;
        }
        r.forced = p.forced_len > 0 && (!p.forced_lens || len < p.forced_lens[b]);
        rs = r;
    }
    __syncthreads();
    const RowState r = rs;
    const int tb = p.timestamp_begin;
    const RowRanges g = row_ranges(p, r);

    // ---- pass 1: masked max / argmax of text (< tb) and timestamp (>= tb) ranges.
    // 16-byte logit loads + 4-byte mask loads, 4 independent groups in flight per thread: only B
    // blocks run, so the pass is latency bound unless the loads are batched
    float mt = -INFINITY, ms = -INFINITY;
    int it = 0x7fffffff, is = 0x7fffffff;
    unsigned okm[RV];       // bit j: logit 4q+j takes part (not suppressed)
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int q = q_lo + tid + 1024 * i, v0 = 4 * q;
            const unsigned char mm[4] = {m4[i].x, m4[i].y, m4[i].z, m4[i].w};
            const bool clean = q < q_hi && !(mm[0] | mm[1] | mm[2] | mm[3]) &&
                               (unsigned)(g.ban0 - v0) > 3u && (unsigned)(g.ban1 - v0) > 3u && (unsigned)(g.ban2 - v0) > 3u;
            const bool all_text = clean && v0 >= g.t_lo && v0 + 3 < g.t_hi;          // t_hi <= tb
            const bool all_ts = clean && v0 >= g.s_lo && v0 + 3 < g.s_hi;            // s_lo >= tb
            unsigned ok = 0;
            if (all_text || all_ts) {
                // the common case: four admissible logits of one class -- a max and, only if it wins, its first position
                ok = 0xFu;
                const float gm = fmaxf(fmaxf(xs[i][0], xs[i][1]), fmaxf(xs[i][2], xs[i][3]));
                const int gj = xs[i][0] == gm ? 0 : xs[i][1] == gm ? 1 : xs[i][2] == gm ? 2 : 3;
                if (all_text) { if (gm > mt) { mt = gm; it = v0 + gj; } }
                else { if (gm > ms) { ms = gm; is = v0 + gj; } }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = v0 + j;
                    if (q < q_hi && !mm[j] && allowed_dyn(g, v)) {
                        ok |= 1u << j;
                        if (v < tb) argmax_merge(mt, it, xs[i][j], v); else argmax_merge(ms, is, xs[i][j], v);
                    }
                }
            }
            okm[i] = ok | (all_text ? 16u : 0u) | (all_ts ? 32u : 0u);
        }
        for (int v = 4 * nvec + tid; kblk == S - 1 && v < p.n_vocab; v += blockDim.x) {
            if (suppressed(p, r, v)) continue;
            if (v < tb) argmax_merge(mt, it, lg[v], v); else argmax_merge(ms, is, lg[v], v);
        }
    } else if (vec_ok) {
#pragma unroll 4
        for (int q = tid; q < nvec; q += blockDim.x) {
            const f32x4 x = lg4[q];
            const uchar4 m4 = sup4[q];
            const unsigned char mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = 4 * q + j;
                if (mm[j] || !allowed_dyn(g, v)) continue;
                if (v < tb) argmax_merge(mt, it, x[j], v); else argmax_merge(ms, is, x[j], v);
            }
        }
        for (int v = 4 * nvec + tid; v < p.n_vocab; v += blockDim.x) {
            if (suppressed(p, r, v)) continue;
            if (v < tb) argmax_merge(mt, it, lg[v], v); else argmax_merge(ms, is, lg[v], v);
        }
    } else {
        for (int v = tid; v < p.n_vocab; v += blockDim.x) {
            if (suppressed(p, r, v)) continue;
            const float x = lg[v];
            if (v < tb) argmax_merge(mt, it, x, v); else argmax_merge(ms, is, x, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        argmax_merge(mt, it, __shfl_xor(mt, o, 64), __shfl_xor(it, o, 64));
        argmax_merge(ms, is, __shfl_xor(ms, o, 64), __shfl_xor(is, o, 64));
    }
    if (lane == 0) {
        sv[0][wave] = mt; si[0][wave] = it;
        sv[1][wave] = ms; si[1][wave] = is;
    }
    __syncthreads();
    mt = sv[0][0]; it = si[0][0]; ms = sv[1][0]; is = si[1][0];
    for (int w = 1; w < 16; ++w) {
        argmax_merge(mt, it, sv[0][w], si[0][w]);
        argmax_merge(ms, is, sv[1][w], si[1][w]);
    }
    const float M = fmaxf(mt, ms);

    // ---- pass 2: exp sums relative to M (a block of a split row may hold no admissible logit at all: M = -inf)
    float st = 0.f, ss = 0.f;
    if (S > 1 && !(M > -INFINITY)) {
    } else if (in_regs) {
#pragma unroll
        for (int i = 0; i < RV; ++i) {
            const int q = q_lo + tid + 1024 * i;
            if (okm[i] & 48u) {
                const float e = (fast_exp(xs[i][0] - M) + fast_exp(xs[i][1] - M)) + (fast_exp(xs[i][2] - M) + fast_exp(xs[i][3] - M));
                if (okm[i] & 16u) st += e; else ss += e;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (okm[i] & (1u << j)) {
                        const float e = fast_exp(xs[i][j] - M);
                        if (4 * q + j < tb) st += e; else ss += e;
                    }
                }
            }
        }
        for (int v = 4 * nvec + tid; kblk == S - 1 && v < p.n_vocab; v += blockDim.x) {
            if (suppressed(p, r, v)) continue;
            const float e = fast_exp(lg[v] - M);
            if (v < tb) st += e; else ss += e;
        }
    } else if (vec_ok) {
#pragma unroll 4
        for (int q = tid; q < nvec; q += blockDim.x) {
            const f32x4 x = lg4[q];
            const uchar4 m4 = sup4[q];
            const unsigned char mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = 4 * q + j;
                if (mm[j] || !allowed_dyn(g, v)) continue;
                const float e = fast_exp(x[j] - M);
                if (v < tb) st += e; else ss += e;
            }
        }
        for (int v = 4 * nvec + tid; v < p.n_vocab; v += blockDim.x) {
            if (suppressed(p, r, v)) continue;
            const float e = fast_exp(lg[v] - M);
            if (v < tb) st += e; else ss += e;
        }
    } else {
        for (int v = tid; v < p.n_vocab; v += blockDim.x) {
            if (suppressed(p, r, v)) continue;
            const float e = fast_exp(lg[v] - M);
            if (v < tb) st += e; else ss += e;
        }
    }
    st = wave_sum(st);
    ss = wave_sum(ss);
    if (lane == 0) {
        ssum[0][wave] = st;
        ssum[1][wave] = ss;
    }
    __syncthreads();
    if constexpr (S > 1) {
        // hand this block's record over; the last block of the row to do so merges them all
        __shared__ int s_last;
        if (tid == 0) {
            st = ss = 0.f;
            for (int w = 0; w < 16; ++w) {
                st += ssum[0][w];
                ss += ssum[1][w];
            }
            float* rec = p.part + ((long)b * S + kblk) * 8;
            rec[0] = mt; rec[1] = __int_as_float(it); rec[2] = ms; rec[3] = __int_as_float(is);
            rec[4] = st; rec[5] = ss; rec[6] = M;
            __threadfence();
            const unsigned t = atomicAdd(p.row_ticket + b, 1u);
            s_last = (t == S - 1);
            if (t == S - 1) p.row_ticket[b] = 0;
        }
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) {
            __threadfence();
            float gmt = -INFINITY, gms = -INFINITY, Mk[S], stk[S], ssk[S];
            int git = 0x7fffffff, gis = 0x7fffffff;
            for (int k = 0; k < S; ++k) {                   // vocabulary order: the first position of a maximum wins
                const float* rec = p.part + ((long)b * S + k) * 8;
                float v[7];
                for (int j = 0; j < 7; ++j) v[j] = __hip_atomic_load(rec + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                argmax_merge(gmt, git, v[0], __float_as_int(v[1]));
                argmax_merge(gms, gis, v[2], __float_as_int(v[3]));
                stk[k] = v[4]; ssk[k] = v[5]; Mk[k] = v[6];
            }
            const float Mg = fmaxf(gmt, gms);
            float gst = 0.f, gss = 0.f;
            for (int k = 0; k < S; ++k)
                if (Mk[k] > -INFINITY) {
                    const float f = expf(Mk[k] - Mg);
                    gst += stk[k] * f;
                    gss += ssk[k] * f;
                }
            sv[0][0] = gmt; si[0][0] = git; sv[1][0] = gms; si[1][0] = gis;
            ssum[0][0] = gst; ssum[1][0] = gss; sv[0][1] = Mg;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float Mfin = M;
        if constexpr (S > 1) {
            mt = sv[0][0]; it = si[0][0]; ms = sv[1][0]; is = si[1][0];
            st = ssum[0][0]; ss = ssum[1][0]; Mfin = sv[0][1];
        } else {
            st = ss = 0.f;
            for (int w = 0; w < 16; ++w) {
                st += ssum[0][w];
                ss += ssum[1][w];
            }
        }
        // timestamp-probability rule: logsumexp(ts) > max(text)  (log-probs share the same lse)
        const bool force_ts = (p.rules & RULE_TS_PROB) && (logf(ss) + Mfin > mt);
        int next;
        float lse, lnext;
        if (force_ts) {
            next = is; lnext = ms; lse = Mfin + logf(ss);
        } else {
            if (mt >= ms) { next = it; lnext = mt; } else { next = is; lnext = ms; }
            lse = Mfin + logf(st + ss);
        }
        if (r.first && p.no_speech_prob) {
            // mlx_whisper_batch_decoder.py:346-352: softmax of the FILTERED logits at no_speech
            float ns = 0.f;
            if (!force_ts && !suppressed(p, r, p.no_speech)) ns = expf(lg[p.no_speech] - lse);
            p.no_speech_prob[b] = ns;
        }
        // a row whose logits are all NaN (poisoned upstream by a bounded wait that gave up; the context's device flag says
        // so) or all masked has no admissible maximum: its position index is the sentinel.  It ends here with EOT --
        // an out-of-range id must never reach the embedding lookup of the next step.
        if ((unsigned)next >= (unsigned)p.n_vocab) next = p.eot;
        if (p.forced_len > 0 && p.forced_lens && n - p.sample_begin >= p.forced_lens[b]) next = p.eot;   // bench workload: this row's length
        if (b >= n_active) next = p.eot;           // a padding row of a pass cut to one launch shape: finished from its first token on
        const int last = tok[n - 1];
        if (last == p.eot) {
            next = p.eot;                          // finished rows keep emitting EOT (:291-293)
        } else {
            p.sum_logprob[b] += lnext - lse;       // (:287-289)
        }
        tok[n] = next;
        // from the next position on this row takes no part in the attention kernels (the reference forwards active
        // sequences only, mlx_whisper_batch_decoder.py:361-373); its later tokens are pinned to EOT above whatever its logits
        if (p.done) p.done[b] = (next == p.eot);
        s_next = next;
    }
    if (p.emb) {
        __syncthreads();
        tail(s_next);
    }
}

__global__ void advance_kernel(int* d_pos, int* d_row, int sample_begin) {
    const int p = *d_pos + 1;
    *d_pos = p;
    *d_row = p - (sample_begin - 1);
}

}  // namespace

hipError_t launch_sample(const SampleArgs& a, hipStream_t s) {
    constexpr int SPLIT = 4, RVS = 4;
    const int nvec = a.n_vocab >> 2;
    const bool vec_ok = ((a.ldl & 3) == 0) && ((reinterpret_cast<size_t>(a.suppress) & 3) == 0) &&
                        ((reinterpret_cast<size_t>(a.logits) & 15) == 0);
    if (a.part && a.row_ticket && vec_ok && nvec >= 4096 && (nvec + SPLIT - 1) / SPLIT <= RVS * 1024) {
        hipLaunchKernelGGL((sample_kernel<SPLIT, RVS>), dim3(a.B, SPLIT), dim3(1024), 0, s, a);
    } else {
        hipLaunchKernelGGL((sample_kernel<1, 13>), dim3(a.B), dim3(1024), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_advance(int* d_pos, int* d_row, int sample_begin, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, s, d_pos, d_row, sample_begin);
    return hipGetLastError();
}
