// Launch-overhead lab (not part of the product): time per kernel of a dependent chain of trivial
// kernels on one stream, eager and as a replayed hipGraph, for a few grid shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

__global__ void k_empty(int* p) {}
__global__ void k_touch(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
__global__ void k_stream(const float4* __restrict__ src, float* __restrict__ dst, int n4_per_block) {
    // each block reads n4_per_block float4 (weights-like stream) and writes one float
    const float4* s = src + (size_t)blockIdx.x * n4_per_block;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n4_per_block; i += blockDim.x) { float4 v = s[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) dst[blockIdx.x] = acc;
}

int main() {
    int* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    float4* src; CK(hipMalloc(&src, 64 << 20)); CK(hipMemset(src, 0, 64 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 512;
    struct Cfg { const char* name; int kind, grid, block; } cfgs[] = {
        {"empty 1x64", 0, 1, 64}, {"empty 256x512", 0, 256, 512}, {"touch 256x512", 1, 256, 512},
        {"touch 80x512", 1, 80, 512}, {"stream 3.3MB 80x512", 2, 80, 512}, {"stream 13MB 320x512", 2, 320, 512}};
    for (auto& c : cfgs) {
        auto launch = [&]() {
            if (c.kind == 0) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, s, d);
            else if (c.kind == 1) hipLaunchKernelGGL(k_touch, dim3(c.grid), dim3(c.block), 0, s, d);
            else hipLaunchKernelGGL(k_stream, dim3(c.grid), dim3(c.block), 0, s, src, (float*)d, 2560);
        };
        for (int i = 0; i < 16; ++i) launch();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const float eager = ms * 1e3f / N;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < N; ++i) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-24s eager %.2f us/kernel   graph %.2f us/kernel\n", c.name, eager, ms * 1e3f / N);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
