// Ring lab (not part of the product): three decode passes in flight, each a chain of latency-bound "GEMV" kernels
// and one HBM-bound "cross-attention" kernel per layer.  (a) the chains run freely on three streams (what bench.py does
// today), (b) the cross-attention kernels of the three passes are ordered in a ring by graph edges, so that exactly one
// of them streams from HBM at any time while the other two passes run their GEMV chains in its shadow.
//   hipcc --offload-arch=gfx950 -O3 tools/ring_lab.hip -o tools/_bin/ring_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

// streams `n16` 16-byte words per block-range with 4 loads in flight per thread; one float out per block
__global__ __launch_bounds__(256) void k_read(const f4* __restrict__ src, size_t n16, float* __restrict__ out) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    float acc = 0.f;
    for (size_t i = lo + threadIdx.x; i < hi; i += 1024) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t j = i + u * 256 < hi ? i + u * 256 : hi - 1;
            v[u] = __builtin_nontemporal_load(src + j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 256; i += 32) t += red[i];
        out[blockIdx.x] = t;
    }
}

int main() {
    const int L = 32, P = 3;
    const size_t MB = 1 << 20;
    const size_t w_small[7] = {(size_t)(9.8 * MB), (size_t)(0.4 * MB), (size_t)(3.3 * MB), (size_t)(3.3 * MB), (size_t)(3.3 * MB), 13 * MB, 13 * MB};
    // order per layer: [LN+QKV, self-attn, out-proj, LN+cross-Q]  CROSS  [out-proj, LN+FC1, FC2]
    const size_t kv_bytes = 123 * MB;
    f4 *weights, *kv[P];
    float* out;
    size_t w_layer = 0;
    for (size_t b : w_small) w_layer += b;
    CK(hipMalloc(&weights, w_layer * L));
    for (int p = 0; p < P; ++p) CK(hipMalloc(&kv[p], kv_bytes * L));
    CK(hipMalloc(&out, 4096 * 4));
    CK(hipMemset(weights, 0, w_layer * L));
    for (int p = 0; p < P; ++p) CK(hipMemset(kv[p], 0, kv_bytes * L));
    hipStream_t s[P];
    for (int p = 0; p < P; ++p) CK(hipStreamCreateWithFlags(&s[p], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(P * L), fork(P), join(P);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : fork) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : join) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto small = [&](int p, int l, int k) {
        size_t off = 0;
        for (int i = 0; i < k; ++i) off += w_small[i];
        const f4* src = reinterpret_cast<const f4*>(reinterpret_cast<const char*>(weights) + (size_t)l * w_layer + off);
        hipLaunchKernelGGL(k_read, dim3(320), dim3(256), 0, s[p], src, w_small[k] / 16, out + p * 1024);
    };
    auto cross = [&](int p, int l) {
        const f4* src = reinterpret_cast<const f4*>(reinterpret_cast<const char*>(kv[p]) + (size_t)l * kv_bytes);
        hipLaunchKernelGGL(k_read, dim3(640), dim3(256), 0, s[p], src, kv_bytes / 16, out + p * 1024);
    };
    for (int mode = 0; mode < 3; ++mode) {     // 0: one pass alone, 1: three passes free-running, 2: three passes, ring
        const int np = mode == 0 ? 1 : P;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork[0], s[0]));
        for (int p = 1; p < np; ++p) CK(hipStreamWaitEvent(s[p], fork[0], 0));
        for (int l = 0; l < L; ++l) {
            for (int p = 0; p < np; ++p)
                for (int k = 0; k < 4; ++k) small(p, l, k);
            for (int p = 0; p < np; ++p) {
                if (mode == 2) {
                    if (p > 0) CK(hipStreamWaitEvent(s[p], ev[(p - 1) * L + l], 0));
                    else if (l > 0) CK(hipStreamWaitEvent(s[p], ev[(np - 1) * L + l - 1], 0));
                }
                cross(p, l);
                if (mode == 2) CK(hipEventRecord(ev[p * L + l], s[p]));
            }
            for (int p = 0; p < np; ++p)
                for (int k = 4; k < 7; ++k) small(p, l, k);
        }
        for (int p = 1; p < np; ++p) {
            CK(hipEventRecord(join[p], s[p]));
            CK(hipStreamWaitEvent(s[0], join[p], 0));
        }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = 20;
        CK(hipEventRecord(e0, s[0]));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s[0]));
        CK(hipEventRecord(e1, s[0]));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_tok = ms * 1e3 / reps, gb = (double)np * L * (kv_bytes + w_layer) / 1e9;
        printf("%-34s %8.1f us per token-step of %d pass(es) = %6.1f us per pass, %5.2f TB/s\n",
               mode == 0 ? "one pass alone" : mode == 1 ? "three passes, free-running" : "three passes, cross-attention ring", us_tok, np,
               us_tok / np, gb / (us_tok * 1e-6) * 1e-3);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
