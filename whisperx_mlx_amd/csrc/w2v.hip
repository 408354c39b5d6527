// wav2vec2 kernels that are not plain GEMM / LayerNorm / attention (SURVEY 8a row 14;
// the forward the reference runs at /root/reference/whisperx/alignment.py:251-258).
//  * conv0 (1 -> C channels, k=10, s=5) fused with GroupNorm(C groups)+GELU: the GroupNorm
//    statistics of every channel over the VALID frames come from 65 sums over the signal
//    (w2v_conv0_sigstats_kernel: double, fixed reduction order), then one pass computes the
//    10-tap convolution, normalises, applies the affine + GELU and writes fp16 channels-last.
//  * row masking of the padded batch before the (zero padded) positional conv.
//  * lm_head + log_softmax fused: one wave per frame, a lane per label.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int C0_FRAMES = 256;   // frames per block

// GroupNorm statistics of conv0 WITHOUT running the convolution over all channels (round 4).  The conv output of channel c at
// frame t is v = sum_k w[c][k] x[5t + k], so over the valid frames
//     sum_t v   = sum_k w[c][k] S[k]                    S[k]     = sum_t x[5t + k]                 (10 numbers)
//     sum_t v^2 = sum_{k,k'} w[c][k] w[c][k'] R[k][k']  R[k][k'] = sum_t x[5t + k] x[5t + k']     (55 numbers, symmetric)
// -- 65 sums over the SIGNAL serve all C channels (the first version ran the 10-tap convolution for every channel a first
// time just to sum it: 1.08 ms of the 3.65 ms conv0 took per 64 x 30 s).  Products of two floats are exact in double and
// the sums are kept in double, reduced in a fixed order (no atomics: the statistics, hence the emissions, are run-to-run
// bit-identical).  C0S_BLK blocks per segment write partial sums [S][C0S_BLK][65] into W2vConv0Args::stats.
constexpr int C0S_BLK = 8, C0S_N = 65;

__global__ __launch_bounds__(256) void w2v_conv0_sigstats_kernel(W2vConv0Args p) {
    __shared__ double red[4][C0S_N];
    const int s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nvalid = p.n_frames[s];
    const float* __restrict__ pcm = p.pcm + (long)s * p.pcm_stride;
    const int per = (nvalid + C0S_BLK - 1) / C0S_BLK;
    const int f0 = blockIdx.x * per, f1 = min(nvalid, f0 + per);
    double acc[C0S_N];
#pragma unroll
    for (int i = 0; i < C0S_N; ++i) acc[i] = 0.0;
    for (int t = f0 + tid; t < f1; t += 256) {
        double x[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const long idx = (long)t * 5 + k;
            x[k] = (idx < p.pcm_stride) ? (double)pcm[idx] : 0.0;
        }
        int i = 10;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            acc[k] += x[k];
#pragma unroll
            for (int k2 = k; k2 < 10; ++k2) acc[i++] += x[k] * x[k2];
        }
    }
#pragma unroll
    for (int i = 0; i < C0S_N; ++i) {
        double v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (tid < C0S_N) p.stats[((long)s * C0S_BLK + blockIdx.x) * C0S_N + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// conv0 -> GroupNorm (statistics from w2v_conv0_sigstats_kernel) -> affine -> GELU, fp16 channels-last
__global__ __launch_bounds__(256) void w2v_conv0_apply_sig_kernel(W2vConv0Args p) {
    __shared__ float xs[C0_FRAMES * 5 + 16];
    __shared__ double sig[C0S_N];
    const int s = blockIdx.y, t0 = blockIdx.x * C0_FRAMES, tid = threadIdx.x;
    const int nvalid = p.n_frames[s];
    const int nt = min(C0_FRAMES, p.Tmax - t0);
    if (nt <= 0) return;
    const float* __restrict__ pcm = p.pcm + (long)s * p.pcm_stride;
    const int nload = nt * p.stride + p.kernel - p.stride;
    for (int i = tid; i < nload; i += 256) {
        const long idx = (long)t0 * p.stride + i;
        xs[i] = (idx < p.pcm_stride) ? pcm[idx] : 0.f;
    }
    if (tid < C0S_N) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < C0S_BLK; ++b) v += p.stats[((long)s * C0S_BLK + b) * C0S_N + tid];
        sig[tid] = v;
    }
    __syncthreads();
    const double n = (double)nvalid;
    for (int c = tid; c < p.C; c += 256) {
        float w[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) w[k] = p.w[c * 10 + k];
        double m1 = 0.0, m2 = 0.0;
        {
            // (rolled on purpose: unrolled, the 65 sums sit in 130 registers for the whole frame loop below and the kernel
            // runs at a quarter of the occupancy)
            const float* wc = p.w + c * 10;
            int i = 10;
#pragma unroll 1
            for (int k = 0; k < 10; ++k) {
                const double wk = (double)wc[k];
                m1 += wk * sig[k];
#pragma unroll 1
                for (int k2 = k; k2 < 10; ++k2) {
                    const double ww = wk * (double)wc[k2];
                    m2 += (k2 == k ? ww : 2.0 * ww) * sig[i++];
                }
            }
        }
        const double mean = m1 / n;
        const double var = m2 / n - mean * mean;
        const float rstd = (float)(1.0 / sqrt(fmax(var, 0.0) + 1e-5));
        const float mu = (float)mean, g = (float)p.gamma[c], b = (float)p.beta[c];
        h16* out = p.out + ((long)s * p.Tmax + t0) * p.C + c;
        for (int t = 0; t < nt; ++t) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 10; ++k) v = fmaf(w[k], xs[t * 5 + k], v);
            out[(long)t * p.C] = (h16)gelu_f((v - mu) * rstd * g + b);
        }
    }
}

// conv0 (+bias) -> LayerNorm over the C channels of every frame -> GELU  (feat_extract_norm = "layer")
constexpr int C0L_FRAMES = 8;
__global__ __launch_bounds__(256) void w2v_conv0_ln_kernel(W2vConv0Args p, const h16* __restrict__ bias) {
    __shared__ float xs[C0L_FRAMES * 5 + 16];
    __shared__ float red[C0L_FRAMES][4];
    const int s = blockIdx.y, t0 = blockIdx.x * C0L_FRAMES, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = min(C0L_FRAMES, p.Tmax - t0);
    if (nt <= 0) return;
    const float* __restrict__ pcm = p.pcm + (long)s * p.pcm_stride;
    for (int i = tid; i < C0L_FRAMES * 5 + 5; i += 256) {
        const long idx = (long)t0 * 5 + i;
        xs[i] = (idx < p.pcm_stride) ? pcm[idx] : 0.f;
    }
    __syncthreads();
    constexpr int MAXC = 4;                 // C <= 1024
    float v[MAXC][C0L_FRAMES];
#pragma unroll
    for (int e = 0; e < MAXC; ++e) {
        const int c = tid + 256 * e;
        const bool ok = c < p.C;
        float w[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) w[k] = ok ? p.w[c * 10 + k] : 0.f;
        const float bc = ok ? (float)bias[c] : 0.f;
#pragma unroll
        for (int f = 0; f < C0L_FRAMES; ++f) {
            float a = bc;
#pragma unroll
            for (int k = 0; k < 10; ++k) a = fmaf(w[k], xs[f * 5 + k], a);
            v[e][f] = ok ? a : 0.f;
        }
    }
    // per-frame mean over channels
    float mean[C0L_FRAMES], rstd[C0L_FRAMES];
#pragma unroll
    for (int f = 0; f < C0L_FRAMES; ++f) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < MAXC; ++e) sum += v[e][f];
        sum = wave_sum(sum);
        if (lane == 0) red[f][wave] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < C0L_FRAMES; ++f) mean[f] = (red[f][0] + red[f][1] + red[f][2] + red[f][3]) / (float)p.C;
    __syncthreads();
#pragma unroll
    for (int f = 0; f < C0L_FRAMES; ++f) {
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < MAXC; ++e) {
            const float d = (tid + 256 * e < p.C) ? v[e][f] - mean[f] : 0.f;
            q += d * d;
        }
        q = wave_sum(q);
        if (lane == 0) red[f][wave] = q;
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < C0L_FRAMES; ++f) rstd[f] = rsqrtf((red[f][0] + red[f][1] + red[f][2] + red[f][3]) / (float)p.C + 1e-5f);
#pragma unroll
    for (int e = 0; e < MAXC; ++e) {
        const int c = tid + 256 * e;
        if (c >= p.C) continue;
        const float g = (float)p.gamma[c], b = (float)p.beta[c];
        h16* out = p.out + ((long)s * p.Tmax + t0) * p.C + c;
#pragma unroll
        for (int f = 0; f < C0L_FRAMES; ++f)
            if (f < nt) out[(long)f * p.C] = (h16)gelu_f((v[e][f] - mean[f]) * rstd[f] * g + b);
    }
}

__global__ void w2v_mask_rows_kernel(h16* x, long seg_stride, long row0, int Tmax, int d, const int* __restrict__ lens) {
    const int s = blockIdx.y;
    const int len = lens[s];
    const long n = (long)(Tmax - len) * (d >> 3);
    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    h16* base = x + (long)s * seg_stride + (row0 + len) * (long)d;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        *reinterpret_cast<half8*>(base + i * 8) = z;
}

__global__ __launch_bounds__(256) void w2v_lmhead_kernel(const h16* __restrict__ x, const h16* __restrict__ w,
                                                         const h16* __restrict__ bias, float* __restrict__ logp,
                                                         int rows, int d, int V, long x_seg, long logp_seg) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    x += blockIdx.y * x_seg;            // blockIdx.y = segment: one launch for the whole padded batch
    logp += blockIdx.y * logp_seg;
    float acc = -INFINITY;
    if (lane < V) {
        const h16* xr = x + (long)row * d;
        const h16* wr = w + (long)lane * d;
        float a = 0.f;
        for (int c = 0; c < d; c += 8) {
            const half8 xv = *reinterpret_cast<const half8*>(xr + c);
            const half8 wv = *reinterpret_cast<const half8*>(wr + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf((float)xv[j], (float)wv[j], a);
        }
        acc = a + (float)bias[lane];
    }
    const float m = wave_max(acc);
    const float e = (lane < V) ? expf(acc - m) : 0.f;
    const float lse = m + logf(wave_sum(e));
    if (lane < V) logp[(long)row * V + lane] = acc - lse;
}

// lm_head + log_softmax on the matrix cores (round 4; V <= 32, d % 16 == 0: the wav2vec2-base / large CTC heads).  The lane-per-
// label kernel above walks d with 2 x 8 conversions and 8 FMAs per step on half-empty waves: 0.9 ms for 64 x 30 s, 30x what
// the 147 MB it reads cost.  Here a wave takes 32 frames: logits^T[label][frame] = W[label][:] . x[frame][:] as 32x32x16 MFMAs
// (A = the label rows of W, zero-clamped beyond V; B = the frames), so a lane ends with 16 of its frame's 32 logits and its
// partner lane (+32) with the other 16: max / sum over 16 registers and one cross-half shuffle, no LDS.
__global__ __launch_bounds__(256) void w2v_lmhead_mfma_kernel(const h16* __restrict__ x, const h16* __restrict__ w,
                                                              const h16* __restrict__ bias, float* __restrict__ logp,
                                                              int rows, int d, int V, long x_seg, long logp_seg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    x += blockIdx.y * x_seg;
    logp += blockIdx.y * logp_seg;
    const int frame = blockIdx.x * 128 + wave * 32 + lr;
    if (blockIdx.x * 128 + wave * 32 >= rows) return;      // (wave-uniform)
    const h16* __restrict__ xr = x + (long)min(frame, rows - 1) * d + 8 * lh;
    const h16* __restrict__ wr = w + (long)min(lr, V - 1) * d + 8 * lh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ns = d >> 4;
#pragma unroll 4
    for (int st = 0; st < ns; ++st) {
        const half8 a = *reinterpret_cast<const half8*>(wr + 16 * st);
        const half8 b = *reinterpret_cast<const half8*>(xr + 16 * st);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int v = (r & 3) + 8 * (r >> 2) + 4 * lh;
        acc[r] = (v < V) ? acc[r] + (float)bias[min(v, V - 1)] : -INFINITY;
        m = fmaxf(m, acc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float e = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) e += expf(acc[r] - m);
    e += __shfl_xor(e, 32, 64);
    const float lse = m + logf(e);
    if (frame < rows) {
        float* __restrict__ o = logp + (long)frame * V;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = 8 * g + 4 * lh + i;
                if (v < V) o[v] = acc[4 * g + i] - lse;
            }
    }
}

// general vocabulary (character sets of e.g. the ja / zh align models have thousands of labels):
// one block per frame, logits staged in LDS, block-wide logsumexp
__global__ __launch_bounds__(256) void w2v_lmhead_big_kernel(const h16* __restrict__ x, const h16* __restrict__ w,
                                                             const h16* __restrict__ bias, float* __restrict__ logp,
                                                             int d, int V, long x_seg, long logp_seg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    x += blockIdx.y * x_seg;
    logp += blockIdx.y * logp_seg;
    float* lg = reinterpret_cast<float*>(smem);            // [V]
    h16* xs = reinterpret_cast<h16*>(smem + (size_t)V * 4);  // [d]
    __shared__ float red[8];
    const long row = blockIdx.x;
    const int tid = threadIdx.x;
    for (int c = tid; c < (d >> 3); c += 256)
        *reinterpret_cast<half8*>(xs + c * 8) = *reinterpret_cast<const half8*>(x + row * d + c * 8);
    __syncthreads();
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) {
        const h16* wr = w + (long)v * d;
        float a = 0.f;
        for (int c = 0; c < d; c += 8) {
            const half8 xv = *reinterpret_cast<const half8*>(xs + c);
            const half8 wv = *reinterpret_cast<const half8*>(wr + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) a = fmaf((float)xv[j], (float)wv[j], a);
        }
        a += (float)bias[v];
        lg[v] = a;
        mx = fmaxf(mx, a);
    }
    mx = block_max(mx, red);
    float se = 0.f;
    for (int v = tid; v < V; v += 256) se += expf(lg[v] - mx);
    se = block_sum(se, red);
    const float lse = mx + logf(se);
    for (int v = tid; v < V; v += 256) logp[row * V + v] = lg[v] - lse;
}

}  // namespace

hipError_t launch_w2v_conv0(const W2vConv0Args& a, int S, hipStream_t s) {
    if (a.kernel != 10 || a.stride != 5) return hipErrorInvalidValue;
    dim3 grid((a.Tmax + C0_FRAMES - 1) / C0_FRAMES, S);
    static_assert(C0S_BLK * C0S_N == W2V_CONV0_STATS_PER_SEGMENT, "statistics buffer layout");
    hipLaunchKernelGGL(w2v_conv0_sigstats_kernel, dim3(C0S_BLK, S), dim3(256), 0, s, a);
    hipLaunchKernelGGL(w2v_conv0_apply_sig_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_w2v_conv0_ln(const W2vConv0Args& a, const h16* bias, int S, hipStream_t s) {
    if (a.kernel != 10 || a.stride != 5 || a.C > 1024) return hipErrorInvalidValue;
    dim3 grid((a.Tmax + C0L_FRAMES - 1) / C0L_FRAMES, S);
    hipLaunchKernelGGL(w2v_conv0_ln_kernel, grid, dim3(256), 0, s, a, bias);
    return hipGetLastError();
}

hipError_t launch_w2v_mask_rows(h16* x, long seg_stride, long row0, int Tmax, int d, const int* lens, int S, hipStream_t s) {
    hipLaunchKernelGGL(w2v_mask_rows_kernel, dim3(32, S), dim3(256), 0, s, x, seg_stride, row0, Tmax, d, lens);
    return hipGetLastError();
}

hipError_t launch_w2v_lmhead(const h16* x, long x_seg, const h16* w, const h16* bias, float* logp, long logp_seg, int S, int rows,
                             int d, int V, hipStream_t s) {
    // S segments of `rows` frames each, x_seg / logp_seg elements apart (the padded batch of wx_w2v_emissions)
    if ((d & 7) || S < 1 || S > 65535) return hipErrorInvalidValue;
    if (V > 64) {
        const size_t lds = (size_t)V * 4 + (size_t)d * 2;
        if (lds > 150 * 1024) return hipErrorInvalidValue;
        hipLaunchKernelGGL(w2v_lmhead_big_kernel, dim3(rows, S), dim3(256), lds, s, x, w, bias, logp, d, V, x_seg, logp_seg);
        return hipGetLastError();
    }
    if (V <= 32 && (d & 15) == 0) {
        hipLaunchKernelGGL(w2v_lmhead_mfma_kernel, dim3((rows + 127) / 128, S), dim3(256), 0, s, x, w, bias, logp, rows, d, V, x_seg, logp_seg);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(w2v_lmhead_kernel, dim3((rows + 3) / 4, S), dim3(256), 0, s, x, w, bias, logp, rows, d, V, x_seg, logp_seg);
    return hipGetLastError();
}
