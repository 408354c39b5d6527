// Prefetch lab (not part of the product): does a load issued by kernel A leave the line where the dependent
// kernel B finds it faster (same-XCD L2 or MALL), across a stream-ordered kernel boundary?
//   chain per iteration:  A(slice i+1 prefetch or nothing)  ->  B(reads slice i+1 ... measured as a whole)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

// B: GEMV-like: block b streams its 40 KB piece of the slice (all loads up front), reduces, writes one float
__global__ __launch_bounds__(512) void k_consume(const float4* __restrict__ slice, float* __restrict__ out, int n4_per_block) {
    const float4* s = slice + (size_t)blockIdx.x * n4_per_block;
    float4 v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) v[u] = s[threadIdx.x + u * 512];
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    __shared__ float red[512];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 512; i += 64) t += red[i];
        out[blockIdx.x] = t;
    }
}
// A: does a little work of its own (reads `own`), optionally with extra blocks that touch the next slice
__global__ __launch_bounds__(512) void k_produce(const float4* __restrict__ own, const float4* __restrict__ next, float* __restrict__ out,
                                                 int n4_per_block, int n_main, int mode) {
    if ((int)blockIdx.x >= n_main) {          // prefetch blocks: block j touches what consumer block j will read (same XCD: j % 8)
        const int j = blockIdx.x - n_main;
        const float4* s = next + (size_t)j * n4_per_block;
        float acc = 0.f;
        if (mode == 1) {                      // one 16-B load per 128-B line is enough to bring the line in
            for (int i = threadIdx.x; i < n4_per_block / 8; i += 512) acc += s[i * 8].x;
        } else {
#pragma unroll
            for (int u = 0; u < 5; ++u) acc += s[threadIdx.x + u * 512].x;
        }
        if (acc == 123.456f) out[blockIdx.x] = acc;
        return;
    }
    const float4* s = own + (size_t)blockIdx.x * n4_per_block;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u) acc += s[threadIdx.x + u * 512].x;
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main() {
    const int NB = 256, n4 = 2560;                         // 256 blocks x 40 KB = 10.5 MB per slice
    const size_t slice4 = (size_t)NB * n4, nsl = 300;      // 3.1 GB: slices are cold when their turn comes
    float4* buf; CK(hipMalloc(&buf, slice4 * nsl * 16)); CK(hipMemset(buf, 0, slice4 * nsl * 16));
    float* out; CK(hipMalloc(&out, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {                 // 0: no prefetch blocks, 1: touch one load per line, 2: full loads
        const int extra = mode == 0 ? 0 : NB;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i + 2 < (int)nsl; i += 2) {
            hipLaunchKernelGGL(k_produce, dim3(NB + extra), dim3(512), 0, s, buf + (size_t)i * slice4, buf + (size_t)(i + 1) * slice4, out, n4, NB, mode);
            hipLaunchKernelGGL(k_consume, dim3(NB), dim3(512), 0, s, buf + (size_t)(i + 1) * slice4, out, n4);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d: %.2f us per (produce, consume) pair\n", mode, ms * 1e3 / ((nsl - 2) / 2));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
