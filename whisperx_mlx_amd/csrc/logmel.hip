// Batched log-mel spectrogram, fp32 throughout.
// Mirrors /root/reference/whisperx/audio.py:112-159 applied per 30 s chunk
// (pad_or_trim to 480000 samples, audio.py:68-91, folded into the bounds):
//   Hann(400, periodic) * frame -> 400-point real DFT -> |.|^2 -> mel filterbank
//   -> log10(clamp 1e-10) -> max(x, chunk_max - 8) -> (x + 4) / 4.
// torch.stft(center=True) reflect-pads 200 samples at both ends of the padded
// 480000-sample chunk; the last (3001st) frame is dropped (audio.py:151).
//
// Kernel 1: one block = 8 consecutive frames of one chunk.  Windowed frames sit in
// LDS; thread k (< 201) accumulates DFT bin k of all 8 frames with a 400-entry
// cos/sin table in LDS indexed by (k*n mod 400), so each twiddle read is reused 8x.
// The sparse triangular mel rows are applied from [lo, lo+len) ranges.  The
// per-chunk max goes through an ordered-uint atomicMax.
// Kernel 2: clamp + scale, writes fp32 (B,3000,n_mels) and/or the fp16 padded
// conv-stem input (B, 3002, c_pad) with one zero row either side.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NBIN = 201, NSAMP = 480000, NFRAMES = 3000, FPB = 8;

__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void logmel_kernel(LogmelArgs p) {
    __shared__ float fr[FPB][NFFT];
    __shared__ float tw[NFFT][2];
    __shared__ float pw[FPB][NBIN + 3];
    __shared__ float red[8];
    const int b = blockIdx.y, f0 = blockIdx.x * FPB, tid = threadIdx.x;
    const float* __restrict__ pcm = p.pcm + (long)b * p.pcm_stride;
    const int nv = min(p.n_valid[b], NSAMP);

    for (int i = tid; i < NFFT; i += 256) {
        tw[i][0] = p.twiddle[2 * i];
        tw[i][1] = p.twiddle[2 * i + 1];
    }
    for (int i = tid; i < FPB * NFFT; i += 256) {
        const int f = i / NFFT, n = i - f * NFFT;
        int s = (f0 + f) * HOP - NFFT / 2 + n;
        if (s < 0) s = -s;                               // reflect at 0
        if (s >= NSAMP) s = 2 * (NSAMP - 1) - s;         // reflect at the padded end
        const float x = (s < nv) ? pcm[s] : 0.f;
        fr[f][n] = x * p.window[n];
    }
    __syncthreads();

    if (tid < NBIN) {
        float re[FPB], im[FPB];
#pragma unroll
        for (int f = 0; f < FPB; ++f) re[f] = im[f] = 0.f;
        int idx = 0;
        for (int n = 0; n < NFFT; ++n) {
            const float c = tw[idx][0], s = tw[idx][1];
#pragma unroll
            for (int f = 0; f < FPB; ++f) {
                const float x = fr[f][n];
                re[f] = fmaf(x, c, re[f]);
                im[f] = fmaf(x, s, im[f]);
            }
            idx += tid;
            if (idx >= NFFT) idx -= NFFT;
        }
#pragma unroll
        for (int f = 0; f < FPB; ++f) pw[f][tid] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();

    float lmax = -INFINITY;
    for (int i = tid; i < FPB * p.n_mels; i += 256) {
        const int f = i / p.n_mels, m = i - f * p.n_mels;
        const int lo = p.filt_lo[m], len = p.filt_len[m];
        const float* __restrict__ w = p.filters + m * NBIN + lo;
        float acc = 0.f;
        for (int k = 0; k < len; ++k) acc = fmaf(w[k], pw[f][lo + k], acc);
        const float v = log10f(fmaxf(acc, 1e-10f));
        p.logspec[((long)b * NFRAMES + f0 + f) * p.n_mels + m] = v;
        lmax = fmaxf(lmax, v);
    }
    lmax = block_max(lmax, red);
    if (tid == 0) atomicMax(p.chunk_max + b, f2ord(lmax));
}

__global__ void logmel_finalize_kernel(const float* __restrict__ logspec, const unsigned* __restrict__ chunk_max,
                                       float* out_f32, h16* out_h, int out_h_ld, int out_h_rows, int n_mels) {
    const int b = blockIdx.y;
    const long n = (long)NFRAMES * n_mels;
    const float floor_ = ord2f(chunk_max[b]) - 8.0f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = (fmaxf(logspec[b * n + i], floor_) + 4.0f) / 4.0f;
        if (out_f32) out_f32[b * n + i] = v;
        if (out_h) {
            const int f = (int)(i / n_mels), m = (int)(i - (long)f * n_mels);
            out_h[((long)b * out_h_rows + f + 1) * out_h_ld + m] = (h16)v;
        }
    }
}

}  // namespace

hipError_t launch_logmel(const LogmelArgs& a, hipStream_t s) {
    hipError_t e = hipMemsetAsync(a.chunk_max, 0, sizeof(unsigned) * a.B, s);
    if (e != hipSuccess) return e;
    dim3 grid(NFRAMES / FPB, a.B);
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_logmel_finalize(const float* logspec, const unsigned* chunk_max, float* out_f32, h16* out_h,
                                  int out_h_ld, int out_h_rows, int B, int n_mels, hipStream_t s) {
    dim3 grid(256, B);
    hipLaunchKernelGGL(logmel_finalize_kernel, grid, dim3(256), 0, s, logspec, chunk_max, out_f32, out_h,
                       out_h_ld, out_h_rows, n_mels);
    return hipGetLastError();
}
