// HBM streaming lab 2 (not part of the product): the decode cross-attention access pattern without the
// attention: grid (H, B, split) x threads, every block reads its K rows then / together with its V rows.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16;
typedef h16 half8 __attribute__((ext_vector_type(8)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int U, int MODE>   // MODE 0: K and V interleaved, 1: K pass then V pass, 2: one wave = one contiguous chunk, 3: as 0 with nt loads, 4: LDS-DMA (nt) into an LDS ring
__global__ void k_cross(const h16* __restrict__ K, const h16* __restrict__ V, float* __restrict__ dst, int T, int nsplit) {
    const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z, tid = threadIdx.x, H = gridDim.x;
    const int per = (((T + nsplit - 1) / nsplit) + 7) & ~7;
    const int k0 = sp * per, k1 = min(T, k0 + per), nkeys = k1 - k0;
    const h16* kp = K + ((long)b * H + h) * T * 64;
    const h16* vp = V + ((long)b * H + h) * T * 64;
    const int lane = tid & 63, wave = tid >> 6, nwave = blockDim.x >> 6, ks = lane >> 3, dc = lane & 7;
    float acc = 0.f;
    if (MODE == 4) {
        // every wave owns a 2-slot ring of U x 2 KiB (K and V pieces of 8 keys each); a slot is refilled right after it is read
        extern __shared__ __attribute__((aligned(16))) char ring[];
        char* mine = ring + wave * (2 * U * 2048);
        const int niter = (nkeys + 8 * nwave - 1) / (8 * nwave);
        auto fill = [&](int it, int slot) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long row = k0 + min(((it + u) * nwave + wave) * 8 + ks, nkeys - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp + row * 64 + dc * 8),
                                                 (__attribute__((address_space(3))) void*)(mine + slot * U * 2048 + u * 2048), 16, 0, 2);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp + row * 64 + dc * 8),
                                                 (__attribute__((address_space(3))) void*)(mine + slot * U * 2048 + u * 2048 + 1024), 16, 0, 2);
            }
        };
        fill(0, 0);
        int slot = 0;
        for (int it = 0; it < niter; it += U) {
            if (it + U < niter) fill(it + U, slot ^ 1);
            if (it + U < niter) { if (U == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const half8 a = *reinterpret_cast<const half8*>(mine + slot * U * 2048 + u * 2048 + lane * 16);
                const half8 b = *reinterpret_cast<const half8*>(mine + slot * U * 2048 + u * 2048 + 1024 + lane * 16);
                acc += (float)a[0] + (float)b[3];
            }
            slot ^= 1;
        }
    } else if (MODE == 2) {
        const int per_w = (nkeys + nwave - 1) / nwave;           // contiguous keys per wave
        const int w0 = wave * per_w, w1 = min(nkeys, w0 + per_w);
        for (int k = w0; k < w1; k += 8 * U) {
            half8 kh[U], vh[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long row = k0 + min(k + u * 8 + ks, nkeys - 1);
                kh[u] = *reinterpret_cast<const half8*>(kp + row * 64 + dc * 8);
                vh[u] = *reinterpret_cast<const half8*>(vp + row * 64 + dc * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += (float)kh[u][0] + (float)vh[u][3];
        }
    } else {
        const int niter = (nkeys + 8 * nwave - 1) / (8 * nwave);
        for (int pass = 0; pass < (MODE == 1 ? 2 : 1); ++pass)
            for (int it = 0; it < niter; it += U) {
                half8 kh[U], vh[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const long row = k0 + min(((it + u) * nwave + wave) * 8 + ks, nkeys - 1);
                    if (MODE == 3) {
                        kh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(kp + row * 64 + dc * 8));
                        vh[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vp + row * 64 + dc * 8));
                    } else {
                    if (MODE == 0 || pass == 0) kh[u] = *reinterpret_cast<const half8*>(kp + row * 64 + dc * 8);
                    if (MODE == 0 || pass == 1) vh[u] = *reinterpret_cast<const half8*>(vp + row * 64 + dc * 8);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (MODE == 0 || MODE == 3 || pass == 0) acc += (float)kh[u][0];
                    if (MODE == 0 || MODE == 3 || pass == 1) acc += (float)vh[u][3];
                }
            }
    }
    if (acc == 123.456f) dst[blockIdx.x] = acc;
}

int main() {
    const int B = 16, H = 20, T = 1500, L = 32;
    const size_t panel = (size_t)B * H * T * 64;          // elements of K (or V) per layer
    h16* kv; CK(hipMalloc(&kv, panel * 2 * L * 2)); CK(hipMemset(kv, 0, panel * 2 * L * 2));
    float* dst; CK(hipMalloc(&dst, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int mode, split, threads, u; } cfgs[] = {{0, 4, 256, 4}, {3, 4, 256, 4}, {3, 2, 256, 4}, {3, 4, 256, 8}, {3, 2, 512, 4},
                                                          {4, 4, 256, 4}, {4, 2, 256, 4}, {4, 4, 256, 8}, {4, 2, 512, 4}, {4, 4, 512, 4}, {4, 8, 256, 4}};
    for (auto& c : cfgs) {
        auto launch = [&](int l) {
            const h16* K = kv + (size_t)l * panel * 2;
            const h16* V = K + panel;
            dim3 g(H, B, c.split), b(c.threads);
            const size_t lds = c.mode == 4 ? (size_t)(c.threads / 64) * 2 * c.u * 2048 : 0;
#define L_(U, M) hipLaunchKernelGGL((k_cross<U, M>), g, b, lds, s, K, V, dst, T, c.split)
            if (c.u == 4) { if (c.mode == 0) L_(4, 0); else if (c.mode == 3) L_(4, 3); else L_(4, 4); }
            else { if (c.mode == 0) L_(8, 0); else if (c.mode == 3) L_(8, 3); else L_(8, 4); }
        };
        for (int l = 0; l < L; ++l) launch(l);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 4; ++r) for (int l = 0; l < L; ++l) launch(l);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (4 * L);
        printf("mode %d split %d threads %4d U=%d   %.2f us/launch   %.0f GB/s\n", c.mode, c.split, c.threads, c.u, us, panel * 4 / us * 1e-3);
    }
    return 0;
}
