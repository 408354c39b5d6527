// Row-wise LayerNorm (fp16 in/out, fp32 statistics, eps 1e-5) and the decoder's
// token + positional embedding gather.  Both are pure HBM streaming: one wave per
// row, 16-byte loads/stores, two-pass variance from registers.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int LN_MAXC = 4;   // up to 4 * 64 * 8 = 2048 columns per row

__global__ __launch_bounds__(256) void layernorm_kernel(const h16* __restrict__ x, long ldx,
                                                        const h16* __restrict__ g, const h16* __restrict__ b,
                                                        h16* __restrict__ y, long ldy, int rows, int d, int gelu) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nchunk = d >> 3;
    const h16* xr = x + (long)row * ldx;
    half8 v[LN_MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            v[c] = *reinterpret_cast<const half8*>(xr + ch * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)v[c][j];
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = (float)v[c][j] - mean;
                q += t * t;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    h16* yr = y + (long)row * ldy;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
            const half8 gg = *reinterpret_cast<const half8*>(g + ch * 8);
            const half8 bb = *reinterpret_cast<const half8*>(b + ch * 8);
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = ((float)v[c][j] - mean) * rstd * (float)gg[j] + (float)bb[j];
                if (gelu) t = gelu_f(t);
                o[j] = (h16)t;
            }
            *reinterpret_cast<half8*>(yr + ch * 8) = o;
        }
    }
}

__global__ void embed_kernel(const int* __restrict__ tokens, int tok_ld, const int* __restrict__ d_pos,
                             const h16* __restrict__ emb, const h16* __restrict__ pos, h16* __restrict__ x, int d) {
    const int b = blockIdx.x;
    const int p = *d_pos;
    const int tok = tokens[(long)b * tok_ld + p];
    const h16* e = emb + (long)tok * d;
    const h16* pe = pos + (long)p * d;
    for (int c = threadIdx.x; c < (d >> 3); c += blockDim.x) {
        const half8 a = *reinterpret_cast<const half8*>(e + c * 8);
        const half8 q = *reinterpret_cast<const half8*>(pe + c * 8);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (h16)((float)a[j] + (float)q[j]);
        *reinterpret_cast<half8*>(x + (long)b * d + c * 8) = o;
    }
}

struct IntPack32 {
    int v[32];
};
__global__ void set_ints_kernel(int* __restrict__ dst, IntPack32 vals, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}

}  // namespace

hipError_t launch_set_ints(int* dst, const int* host_vals, int n, hipStream_t s) {
    for (int off = 0; off < n; off += 32) {
        IntPack32 pk{};
        const int m = n - off < 32 ? n - off : 32;
        for (int i = 0; i < m; ++i) pk.v[i] = host_vals[off + i];     // copied into the launch packet NOW
        hipLaunchKernelGGL(set_ints_kernel, dim3(1), dim3(32), 0, s, dst + off, pk, m);
    }
    return hipGetLastError();
}

hipError_t launch_layernorm(const h16* x, long ldx, const h16* g, const h16* b, h16* y, long ldy, int rows, int d,
                            hipStream_t s, int gelu) {
    if ((d & 7) || d > LN_MAXC * 512) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, g, b, y, ldy, rows, d, gelu);
    return hipGetLastError();
}

hipError_t launch_embed(const int* tokens, int tok_ld, const int* d_pos, const h16* emb, const h16* pos, h16* x,
                        int B, int d, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(128), 0, s, tokens, tok_ld, d_pos, emb, pos, x, d);
    return hipGetLastError();
}
