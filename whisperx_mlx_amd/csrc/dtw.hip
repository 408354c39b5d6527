// Cross-attention DTW word timing.
//   extract_words_with_dtw  /root/reference/mlx_whisper_optimized_final.py:128-253
//   median_filter_fixed     /root/reference/median_filter_fix.py:6-21
//   dtw (third-party mlx_whisper.timing.dtw, called at optimized_final.py:201):
//   published Whisper dtw_cpu/backtrace, strict-< tie rule.
// mode 0 (default, published find_alignment): per head softmax over frames, z-norm
//   over the token axis, median-7 over frames, mean over heads, DTW on -matrix.
// mode 1 (the in-repo variant): mean over heads, softmax(10 x), median-7, per-row
//   z-norm (std + 1e-8), DTW on -matrix^T.
// All tiny and latency-bound: one block per sequence, everything stays on the GPU
// so the decode loop's captured scores never cross PCIe.
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "kernels.h"

namespace {

// ---- which decode steps become alignment rows: text tokens up to and incl. the first EOT
__global__ void dtw_rows_kernel(const int* __restrict__ tokens, int tok_ld, int sample_begin, int rows, int n_sampled, int eot,
                                int* __restrict__ rowmap, int* __restrict__ n_rows) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int* t = tokens + (long)b * tok_ld + sample_begin;
    int n = 0;
    for (int s = 0; s < n_sampled; ++s) {
        const int tk = t[s];
        if (tk < eot) {
            rowmap[b * (rows + 1) + n++] = s;
        } else if (tk == eot) {
            rowmap[b * (rows + 1) + n++] = s;
            break;
        }
    }
    n_rows[b] = n;
}

// frames of sequence b that carry audio (published find_alignment crops to num_frames // 2)
__device__ __forceinline__ int frames_of(const DtwArgs& p, int b) {
    return p.n_frames ? max(8, min(p.n_frames[b], p.T)) : p.T;
}

__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ float median7(float a0, float a1, float a2, float a3, float a4, float a5, float a6) {
    float v[7] = {a0, a1, a2, a3, a4, a5, a6};
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 6 - i; ++j) {
            const float lo = fminf(v[j], v[j + 1]), hi = fmaxf(v[j], v[j + 1]);
            v[j] = lo;
            v[j + 1] = hi;
        }
    return v[3];
}

// mode 0, step 1: softmax over frames of row `r` of head `hd` -> w2[b][hd][r][:]
__global__ __launch_bounds__(256) void dtw_softmax_kernel(DtwArgs p, const int* __restrict__ rowmap,
                                                          const int* __restrict__ n_rows) {
    __shared__ float red[8];
    const int r = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
    if (r >= n_rows[b]) return;
    const int Tb = frames_of(p, b);
    const int s = rowmap[b * (p.rows + 1) + r];
    const float* src = p.qk + (((long)b * p.n_cap + hd) * p.rows + s) * p.T;
    float* dst = p.work2 + (((long)b * p.n_cap + hd) * (p.rows + 1) + r) * p.T;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < Tb; i += 256) mx = fmaxf(mx, src[i] * p.qk_scale);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int i = threadIdx.x; i < Tb; i += 256) {
        const float e = expf(src[i] * p.qk_scale - mx);
        dst[i] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    for (int i = threadIdx.x; i < Tb; i += 256) dst[i] = dst[i] / sum;
}

// mode 0, step 2: z-norm over the token axis for every (head, frame)
__global__ void dtw_znorm_tok_kernel(DtwArgs p, const int* __restrict__ n_rows) {
    const int hd = blockIdx.y, b = blockIdx.z;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames_of(p, b)) return;
    const int n = n_rows[b];
    float* base = p.work2 + ((long)b * p.n_cap + hd) * (p.rows + 1) * p.T + f;
    float s = 0.f;
    for (int r = 0; r < n; ++r) s += base[(long)r * p.T];
    const float mean = s / (float)n;
    float q = 0.f;
    for (int r = 0; r < n; ++r) {
        const float t = base[(long)r * p.T] - mean;
        q += t * t;
    }
    const float std_ = sqrtf(q / (float)n);
    for (int r = 0; r < n; ++r) base[(long)r * p.T] = (base[(long)r * p.T] - mean) / std_;
}

// mode 0, step 3: median-7 over frames per head, then mean over heads -> work[b][r][:]
__global__ void dtw_median_mean_kernel(DtwArgs p, const int* __restrict__ n_rows) {
    const int r = blockIdx.y, b = blockIdx.z;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int Tb = frames_of(p, b);
    if (r >= n_rows[b] || f >= Tb) return;
    float acc = 0.f;
    for (int hd = 0; hd < p.n_cap; ++hd) {
        const float* row = p.work2 + (((long)b * p.n_cap + hd) * (p.rows + 1) + r) * p.T;
        acc += median7(row[reflect(f - 3, Tb)], row[reflect(f - 2, Tb)], row[reflect(f - 1, Tb)], row[f],
                       row[reflect(f + 1, Tb)], row[reflect(f + 2, Tb)], row[reflect(f + 3, Tb)]);
    }
    p.work[((long)b * (p.rows + 1) + r) * p.T + f] = acc / (float)p.n_cap;
}

// mode 1: whole row pipeline in one block (optimized_final.py:171-197)
__global__ __launch_bounds__(256) void dtw_inrepo_row_kernel(DtwArgs p, const int* __restrict__ rowmap,
                                                             const int* __restrict__ n_rows) {
    __shared__ float buf[2][1536];
    __shared__ float red[8];
    const int r = blockIdx.x, b = blockIdx.y;
    if (r >= n_rows[b]) return;
    const int Tb = frames_of(p, b);
    const int s = rowmap[b * (p.rows + 1) + r];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < Tb; i += 256) {
        float a = 0.f;
        for (int hd = 0; hd < p.n_cap; ++hd) a += p.qk[(((long)b * p.n_cap + hd) * p.rows + s) * p.T + i];
        a = a / (float)p.n_cap * 10.0f;
        buf[0][i] = a;
        mx = fmaxf(mx, a);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int i = threadIdx.x; i < Tb; i += 256) {
        const float e = expf(buf[0][i] - mx);
        buf[0][i] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    for (int i = threadIdx.x; i < Tb; i += 256) buf[0][i] /= sum;
    __syncthreads();
    float msum = 0.f;
    for (int f = threadIdx.x; f < Tb; f += 256) {
        const float* row = buf[0];
        const float m = median7(row[reflect(f - 3, Tb)], row[reflect(f - 2, Tb)], row[reflect(f - 1, Tb)], row[f],
                                row[reflect(f + 1, Tb)], row[reflect(f + 2, Tb)], row[reflect(f + 3, Tb)]);
        buf[1][f] = m;
        msum += m;
    }
    msum = block_sum(msum, red);
    const float mean = msum / (float)Tb;
    float q = 0.f;
    for (int f = threadIdx.x; f < Tb; f += 256) {
        const float t = buf[1][f] - mean;
        q += t * t;
    }
    q = block_sum(q, red);
    const float std_ = sqrtf(q / (float)Tb) + 1e-8f;
    for (int f = threadIdx.x; f < Tb; f += 256)
        p.work[((long)b * (p.rows + 1) + r) * p.T + f] = (buf[1][f] - mean) / std_;
}

// ---- DTW: cost x[i][j] = -mat (element strides si, sj), N rows x M cols, anti-diagonal wavefront
__global__ __launch_bounds__(1024) void dtw_wavefront_kernel(DtwArgs p, const int* __restrict__ n_rows) {
    __shared__ float diag[3][1537];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nr = n_rows[b];
    const float* mat = p.work + (long)b * (p.rows + 1) * p.T;
    int N, M;
    long si, sj;
    const int Tb = frames_of(p, b);
    if (p.mode == 0) { N = nr; M = Tb; si = p.T; sj = 1; }
    else             { N = Tb; M = nr; si = 1; sj = p.T; }
    int* out_len = p.path_len + b;
    if (nr <= 0) {
        if (tid == 0) *out_len = 0;
        return;
    }
    unsigned char* trace = p.trace + (long)b * p.trace_stride;
    const long tstride = M + 1;
    // diag buffers indexed by i (0..N); d = i + j
    for (int i = tid; i <= N; i += blockDim.x) {
        diag[0][i] = INFINITY;   // d = 0: only (0,0)
        diag[1][i] = INFINITY;   // d = 1: (0,1),(1,0) both inf
        diag[2][i] = INFINITY;
    }
    __syncthreads();
    if (tid == 0) diag[0][0] = 0.f;
    __syncthreads();
    int i2 = 0, i1 = 1, i0 = 2;    // d-2, d-1, d
    for (int d = 2; d <= N + M; ++d) {
        const int lo = max(1, d - M), hi = min(N, d - 1);
        for (int i = lo + tid; i <= hi; i += blockDim.x) {
            const int j = d - i;
            const float c0 = diag[i2][i - 1], c1 = diag[i1][i - 1], c2 = diag[i1][i];
            float c;
            unsigned char t;
            if (c0 < c1 && c0 < c2) { c = c0; t = 0; }
            else if (c1 < c0 && c1 < c2) { c = c1; t = 1; }
            else { c = c2; t = 2; }
            diag[i0][i] = -mat[(long)(i - 1) * si + (long)(j - 1) * sj] + c;
            trace[(long)i * tstride + j] = t;
        }
        // boundary cells of diagonal d: (0,d) and (d,0) are +inf
        if (tid == 0) {
            diag[i0][0] = INFINITY;
            if (d <= N) diag[i0][d] = INFINITY;
        }
        __syncthreads();
        const int tmp = i2; i2 = i1; i1 = i0; i0 = tmp;
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        // backtrace (published `backtrace`): trace[0,:]=2, trace[:,0]=1
        int i = N, j = M, n = 0;
        int* pi = p.path_i + (long)b * p.path_stride;
        int* pj = p.path_j + (long)b * p.path_stride;
        while (i > 0 || j > 0) {
            pi[n] = i - 1;
            pj[n] = j - 1;
            ++n;
            int t;
            if (i == 0) t = 2;
            else if (j == 0) t = 1;
            else t = trace[(long)i * tstride + j];
            if (t == 0) { --i; --j; }
            else if (t == 1) --i;
            else --j;
        }
        *out_len = n;   // path is stored end -> start; the host reverses it
    }
}

// The same wavefront for mode 0 with <= 256 alignment rows (the normal case: <= 225), without a memory round trip per
// diagonal: one thread per row i walks its row left to right (cell (i, d - i) on diagonal d); the costs of the next
// 32 diagonals sit in LDS (chunk[r][s] = cost of row r on diagonal d0 + s, rows padded to 33 words; the following
// chunk is fetched into registers meanwhile), the 2-bit trace codes are packed 16 per word by the row's thread into
// an LDS image of the trace that the backtrace then reads, and the per-diagonal barrier is LDS-only
// (__syncthreads() would also wait for that diagonal's global stores).  Same arithmetic, same tie rule.
constexpr int DTW_CH = 32;
__global__ __launch_bounds__(256) void dtw_wavefront_lds_kernel(DtwArgs p, const int* __restrict__ n_rows) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nr = n_rows[b];
    int* out_len = p.path_len + b;
    if (nr <= 0) {
        if (tid == 0) *out_len = 0;
        return;
    }
    const int Nmax = p.rows + 1, NP = Nmax + 1, WPR = (p.T + 15) >> 4;
    float* diag = dsm;                                    // [3][NP]
    float* chunk = diag + 3 * NP;                         // [2][Nmax][33]
    unsigned* trace_l = reinterpret_cast<unsigned*>(chunk + 2 * Nmax * 33);   // [Nmax][WPR]
    const float* __restrict__ mat = p.work + (long)b * (p.rows + 1) * p.T;
    const int N = nr, M = frames_of(p, b);
    const int n_el = N * DTW_CH;
    float stage[32];
    auto fetch = [&](int c) {                             // costs of diagonals 2 + 32 c ... into registers
        const int d0 = 2 + c * DTW_CH;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int idx = tid + k * 256, r = idx >> 5, jm1 = d0 + (idx & 31) - r - 2;
            stage[k] = (idx < n_el && jm1 >= 0 && jm1 < M) ? mat[(long)r * p.T + jm1] : 0.f;
        }
    };
    auto park = [&](int c) {
        float* dst = chunk + (c & 1) * Nmax * 33;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int idx = tid + k * 256;
            if (idx < n_el) dst[(idx >> 5) * 33 + (idx & 31)] = stage[k];
        }
    };
    fetch(0);
    for (int i = tid; i < 3 * NP; i += 256) diag[i] = INFINITY;
    park(0);
    __syncthreads();
    if (tid == 0) diag[0] = 0.f;                          // cell (0, 0)
    __syncthreads();
    const int i = tid + 1;                                // this thread's row
    const bool active = i <= N;
    unsigned tw = 0;
    float *dg2 = diag, *dg1 = diag + NP, *dg0 = diag + 2 * NP;   // diagonals d-2, d-1, d
    int sidx = 0, c = 0;
    const float* crow = chunk + (i - 1) * 33;             // this row's costs in the current chunk buffer
    unsigned* trow = trace_l + (i - 1) * WPR;
    const int last = N + M;
    for (int d = 2; d <= last; ++d) {
        if (sidx == 0 && 2 + (c + 1) * DTW_CH <= last) fetch(c + 1);
        if (active) {
            const int j = d - i;
            if (j >= 1 && j <= M) {
                const float c0 = dg2[i - 1], c1 = dg1[i - 1], c2 = dg1[i];
                float cm;
                unsigned t;
                if (c0 < c1 && c0 < c2) { cm = c0; t = 0; }
                else if (c1 < c0 && c1 < c2) { cm = c1; t = 1; }
                else { cm = c2; t = 2; }
                dg0[i] = -crow[sidx] + cm;
                const int jm1 = j - 1;
                tw |= t << (2 * (jm1 & 15));
                if ((jm1 & 15) == 15 || j == M) {
                    trow[jm1 >> 4] = tw;
                    tw = 0;
                }
            } else if (j == 0) {
                dg0[i] = INFINITY;                        // cell (i, 0)
            }
        }
        if (tid == 0) dg0[0] = INFINITY;                  // cell (0, d)
        if (sidx == DTW_CH - 1) {
            if (2 + (c + 1) * DTW_CH <= last) park(c + 1);
            sidx = 0;
            ++c;
            crow = chunk + (c & 1) * Nmax * 33 + (i - 1) * 33;
        } else {
            ++sidx;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float* tmp = dg2; dg2 = dg1; dg1 = dg0; dg0 = tmp;
    }
    if (tid != 0) return;
    // backtrace (published `backtrace`): trace[0,:]=2, trace[:,0]=1
    int bi = N, bj = M, n = 0;
    int* pi = p.path_i + (long)b * p.path_stride;
    int* pj = p.path_j + (long)b * p.path_stride;
    while (bi > 0 || bj > 0) {
        pi[n] = bi - 1;
        pj[n] = bj - 1;
        ++n;
        int t;
        if (bi == 0) t = 2;
        else if (bj == 0) t = 1;
        else t = (trace_l[(bi - 1) * WPR + ((bj - 1) >> 4)] >> (2 * ((bj - 1) & 15))) & 3;
        if (t == 0) { --bi; --bj; }
        else if (t == 1) --bi;
        else --bj;
    }
    *out_len = n;   // path is stored end -> start; the host reverses it
}

// test hook (include/wxhip_test.h: wx_median7_rows): the running median the two kernels above apply -- same
// `reflect` and `median7` -- on a plain matrix, so that it can be held against the reference's median_filter_fixed
__global__ void median7_rows_kernel(const float* __restrict__ x, long ldx, int T, float* __restrict__ y, long ldy) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= T) return;
    const float* row = x + (long)blockIdx.y * ldx;
    y[(long)blockIdx.y * ldy + f] = median7(row[reflect(f - 3, T)], row[reflect(f - 2, T)], row[reflect(f - 1, T)], row[f],
                                            row[reflect(f + 1, T)], row[reflect(f + 2, T)], row[reflect(f + 3, T)]);
}

}  // namespace

hipError_t launch_median7_rows(const float* x, long ldx, int rows, int T, float* y, long ldy, hipStream_t s) {
    if (rows < 1 || T < 4) return hipErrorInvalidValue;      // reflect() folds once: the pad (3) must be shorter than the row
    hipLaunchKernelGGL(median7_rows_kernel, dim3((T + 255) / 256, rows), dim3(256), 0, s, x, ldx, T, y, ldy);
    return hipGetLastError();
}

hipError_t launch_dtw(const DtwArgs& a, hipStream_t s) {
    if (a.T > 1536 || a.rows + 1 > 1536) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dtw_rows_kernel, dim3(a.B), dim3(64), 0, s, a.tokens, a.tok_ld, a.sample_begin, a.rows, a.n_sampled, a.eot,
                       a.rowmap, a.n_rows);
    if (a.mode == 0) {
        hipLaunchKernelGGL(dtw_softmax_kernel, dim3(a.rows + 1, a.n_cap, a.B), dim3(256), 0, s, a, a.rowmap, a.n_rows);
        hipLaunchKernelGGL(dtw_znorm_tok_kernel, dim3((a.T + 255) / 256, a.n_cap, a.B), dim3(256), 0, s, a, a.n_rows);
        hipLaunchKernelGGL(dtw_median_mean_kernel, dim3((a.T + 255) / 256, a.rows + 1, a.B), dim3(256), 0, s, a, a.n_rows);
    } else {
        hipLaunchKernelGGL(dtw_inrepo_row_kernel, dim3(a.rows + 1, a.B), dim3(256), 0, s, a, a.rowmap, a.n_rows);
    }
    const int Nmax = a.rows + 1;
    const size_t lds = sizeof(float) * ((size_t)3 * (Nmax + 1) + (size_t)2 * Nmax * 33 + (size_t)Nmax * ((a.T + 15) / 16));
    if (a.mode == 0 && Nmax <= 256 && lds <= 150 * 1024) {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(dtw_wavefront_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        });
        if (attr_err != hipSuccess) return attr_err;
        hipLaunchKernelGGL(dtw_wavefront_lds_kernel, dim3(a.B), dim3(256), lds, s, a, a.n_rows);
    } else {
        hipLaunchKernelGGL(dtw_wavefront_kernel, dim3(a.B), dim3(1024), 0, s, a, a.n_rows);
    }
    return hipGetLastError();
}
