// HBM streaming lab (not part of the product): what plain 16-byte global loads reach on this chip when
// every launch reads a fresh 123 MB slice of a 3.9 GB buffer (the decode cross-attention access pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int U>
__global__ void k_stream(const float4* __restrict__ src, float* __restrict__ dst, long n4_per_block) {
    const float4* s = src + (size_t)blockIdx.x * n4_per_block;
    float acc = 0.f;
    for (long i = threadIdx.x; i < n4_per_block; i += (long)blockDim.x * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s[i + (long)u * blockDim.x];   // n4_per_block is a multiple of blockDim*U
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) dst[blockIdx.x] = acc;
}

int main() {
    const size_t slice = 134217728, nsl = 30;
    float4* src; CK(hipMalloc(&src, slice * nsl)); CK(hipMemset(src, 0, slice * nsl));
    float* dst; CK(hipMalloc(&dst, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int grid, block, u; } cfgs[] = {{1024, 512, 4}, {1024, 512, 8}, {2048, 256, 4}, {2048, 256, 8}, {512, 1024, 4},
                                                 {512, 1024, 8}, {4096, 256, 4}, {8192, 128, 4}, {256, 1024, 8}, {256, 1024, 4}, {2048, 512, 4}, {4096, 512, 4}};
    for (auto& c : cfgs) {
        const long n4 = slice / 16 / c.grid;
        if (n4 % ((long)c.block * c.u)) { printf("skip %dx%d u%d (n4=%ld)\n", c.grid, c.block, c.u, n4); continue; }
        auto launch = [&](int l) {
            const float4* p = src + (size_t)l * (slice / 16);
            if (c.u == 4) hipLaunchKernelGGL(k_stream<4>, dim3(c.grid), dim3(c.block), 0, s, p, dst, n4);
            else hipLaunchKernelGGL(k_stream<8>, dim3(c.grid), dim3(c.block), 0, s, p, dst, n4);
        };
        for (int l = 0; l < 30; ++l) launch(l);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 4; ++r) for (int l = 0; l < 30; ++l) launch(l);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / 120;
        printf("grid %5d x %4d  U=%d   %.2f us/launch   %.0f GB/s\n", c.grid, c.block, c.u, us, slice / us * 1e-3);
    }
    return 0;
}
