// wav2vec2 kernels that are not plain GEMM / LayerNorm / attention (SURVEY 8a row 14;
// the forward the reference runs at /root/reference/whisperx/alignment.py:251-258).
//  * conv0 (1 -> C channels, k=10, s=5) fused with GroupNorm(C groups)+GELU: the
//    convolution is so cheap (10 MACs per output) that it is recomputed instead of
//    stored: pass 1 accumulates per-(segment, channel) sum / sum-of-squares over the
//    VALID frames (fp32 partials per block, float64 atomics), pass 2 recomputes,
//    normalises, applies the affine + GELU and writes fp16 channels-last.
//  * row masking of the padded batch before the (zero padded) positional conv.
//  * lm_head + log_softmax fused: one wave per frame, a lane per label.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int C0_FRAMES = 256;   // frames per block

template <bool APPLY>
__global__ __launch_bounds__(256) void w2v_conv0_kernel(W2vConv0Args p) {
    __shared__ float xs[C0_FRAMES * 5 + 16];
    const int s = blockIdx.y, t0 = blockIdx.x * C0_FRAMES, tid = threadIdx.x;
    const int nvalid = p.n_frames[s];
    const int nt = APPLY ? min(C0_FRAMES, p.Tmax - t0) : min(C0_FRAMES, nvalid - t0);
    if (nt <= 0) return;
    const float* __restrict__ pcm = p.pcm + (long)s * p.pcm_stride;
    const int nload = nt * p.stride + p.kernel - p.stride;
    for (int i = tid; i < nload; i += 256) {
        const long idx = (long)t0 * p.stride + i;
        xs[i] = (idx < p.pcm_stride) ? pcm[idx] : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        float w[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) w[k] = p.w[c * 10 + k];
        if (!APPLY) {
            float sum = 0.f, sq = 0.f;
            for (int t = 0; t < nt; ++t) {
                float v = 0.f;
#pragma unroll
                for (int k = 0; k < 10; ++k) v = fmaf(w[k], xs[t * 5 + k], v);
                sum += v;
                sq = fmaf(v, v, sq);
            }
            atomicAdd(p.stats + ((long)s * p.C + c) * 2, (double)sum);
            atomicAdd(p.stats + ((long)s * p.C + c) * 2 + 1, (double)sq);
        } else {
            const double n = (double)nvalid;
            const double mean = p.stats[((long)s * p.C + c) * 2] / n;
            const double var = p.stats[((long)s * p.C + c) * 2 + 1] / n - mean * mean;
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
                                                         int rows, int d, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
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

}  // namespace

hipError_t launch_w2v_conv0(const W2vConv0Args& a, int S, hipStream_t s) {
    if (a.kernel != 10 || a.stride != 5) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(a.stats, 0, sizeof(double) * 2 * (size_t)S * a.C, s);
    if (e != hipSuccess) return e;
    dim3 grid((a.Tmax + C0_FRAMES - 1) / C0_FRAMES, S);
    hipLaunchKernelGGL(w2v_conv0_kernel<false>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(w2v_conv0_kernel<true>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_w2v_mask_rows(h16* x, long seg_stride, long row0, int Tmax, int d, const int* lens, int S, hipStream_t s) {
    hipLaunchKernelGGL(w2v_mask_rows_kernel, dim3(32, S), dim3(256), 0, s, x, seg_stride, row0, Tmax, d, lens);
    return hipGetLastError();
}

hipError_t launch_w2v_lmhead(const h16* x, const h16* w, const h16* bias, float* logp, int rows, int d, int V, hipStream_t s) {
    if (V > 64 || (d & 7)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(w2v_lmhead_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, bias, logp, rows, d, V);
    return hipGetLastError();
}
