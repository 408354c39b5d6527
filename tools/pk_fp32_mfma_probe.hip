// Stand-alone reproducer: packed-fp32 VALU results change while another wave on the SIMD issues MFMAs.
//
// Background: with two engine contexts in flight the log-mel kernel returned different bits in a few
// hundred values per call whenever the other context ran the 128^2 GEMM (the two share CUs: 22 KiB
// + 2 x 64 KiB of LDS).  Bisecting the GEMM (tools/coresident_probe.hip) left its MFMAs as the
// cause, and the affected DFT bins were always threads 48-63 of each wave.  This file reduces it to
//
//   victim:    256 threads, 22 KiB static LDS, a loop of broadcast LDS reads and fp32 FMAs that the
//              compiler emits as v_pk_fma_f32; no global loads inside the loop.  Variant 2 is the
//              same loop in packed fp16 (v_pk_fma_f16).
//   aggressor: 256 threads, 64 KiB dynamic LDS (so that two of its blocks and one victim block fit a
//              CU), looping over global_load_lds (mode 0), global_load + ds_write (mode 1) or
//              v_mfma_f32_16x16x32_f16 (mode 2).
//
// Each victim run is compared bit for bit with a run of its own on an idle GPU.  Measured on MI355X
// (ROCm 7.2, profiles/r02_pk_fp32_mfma_probe.txt):
//   * beside modes 0 and 1 nothing changes;
//   * beside mode 2 the fp32 victim differs in 11-12 of 12 trials, ONLY in lanes 48-63, by up to
//     ~1 % of the accumulated value;
//   * built with  -Xclang -target-feature -Xclang -packed-fp32-ops  (two v_fma_f32 instead of one
//     v_pk_fma_f32) it never differs.
// The library is therefore built without packed-fp32 ops (whisperx_mlx_amd/build.py).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_probe tools/pk_fp32_mfma_probe.hip && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr int NT = 400, NF = 8;

__global__ __launch_bounds__(256) void victim(const float* __restrict__ table, const float* __restrict__ frames,
                                              float* __restrict__ out, int rounds, int lds_variant) {
    __shared__ float fr[NF][NT];
    __shared__ float tw[NT][2];
    __shared__ float pad[NF][204];
    const int tid = threadIdx.x;
    for (int i = tid; i < NT; i += 256) { tw[i][0] = table[2 * i]; tw[i][1] = table[2 * i + 1]; }
    for (int i = tid; i < NF * NT; i += 256) fr[i / NT][i % NT] = frames[(blockIdx.x % 64) * NF * NT + i];
    for (int i = tid; i < NF * 204; i += 256) pad[i / 204][i % 204] = 0.f;
    __syncthreads();
    float re[NF], im[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) re[f] = im[f] = 0.f;
    for (int r = 0; r < rounds; ++r) {
        int idx = 0;
        for (int n = 0; n < NT; ++n) {
            float c, s;
            if (lds_variant == 0) { c = tw[idx][0]; s = tw[idx][1]; }
            else { c = __sinf(0.001f * idx); s = 1.f - c; }              // no scattered LDS read
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float x = fr[f][n];
                re[f] = fmaf(x, c, re[f]);
                im[f] = fmaf(x, s, im[f]);
            }
            idx += tid;
            if (idx >= NT) idx -= NT;
        }
    }
    float acc = pad[0][tid % 204];
    if (lds_variant == 2) {   // the same loop shape in packed fp16
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 a2[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) a2[f] = (h2){(_Float16)0.f, (_Float16)0.f};
        for (int r = 0; r < rounds; ++r) {
            int idx = 0;
            for (int n = 0; n < NT; ++n) {
                const h2 cs = {(_Float16)(0.001f * tw[idx][0]), (_Float16)(0.001f * tw[idx][1])};
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const _Float16 x = (_Float16)fr[f][n];
                    a2[f] = (h2){x, x} * cs + a2[f] * (h2){(_Float16)0.999f, (_Float16)0.999f};
                }
                idx += tid;
                if (idx >= NT) idx -= NT;
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) acc += (float)a2[f][0] + (float)a2[f][1];
        out[(long)blockIdx.x * 256 + tid] = acc;
        return;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) acc += re[f] * re[f] + im[f] * im[f];
    out[(long)blockIdx.x * 256 + tid] = acc;
}

__global__ __launch_bounds__(256, 2) void aggressor(const char* __restrict__ src, float* __restrict__ sink, int iters, int mode,
                                                    long src_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc = 0.f;
    const long base = ((long)blockIdx.x * 65536) % (src_bytes - 4 * 65536);
    for (int it = 0; it < iters; ++it) {
        const char* s0 = src + base + (long)(it & 3) * 65536;
        if (mode == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                char* dst = smem + wave * 16384 + j * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s0 + wave * 16384 + j * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode == 2) {
            typedef _Float16 half8 __attribute__((ext_vector_type(8)));
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            half8 a, b;
            for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.01f * (lane + q)); b[q] = (_Float16)(0.02f * (lane - q)); }
            f32x4 c[8];
            for (int q = 0; q < 8; ++q) c[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < 64; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[q], 0, 0, 0);
            for (int q = 0; q < 8; ++q) acc += c[q][0] + c[q][3];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(s0 + wave * 16384 + j * 1024 + lane * 16);
                *reinterpret_cast<float4*>(smem + wave * 16384 + j * 1024 + lane * 16) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(smem + ((tid * 16 + j * 8192 + it * 64) & 65535 & ~15));
            acc += v.x + v.y + v.z + v.w;
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int vblocks = 4096, rounds = 4, trials = 12;
    std::vector<float> table(2 * NT), frames(64 * NF * NT);
    srand(7);
    for (auto& v : table) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : frames) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *d_table, *d_frames, *d_out, *d_gold, *d_sink;
    char* d_src;
    const long src_bytes = 256l << 20;
    CK(hipMalloc(&d_table, table.size() * 4));
    CK(hipMalloc(&d_frames, frames.size() * 4));
    CK(hipMalloc(&d_out, (long)vblocks * 256 * 4));
    CK(hipMalloc(&d_gold, (long)vblocks * 256 * 4));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMalloc(&d_src, src_bytes));
    CK(hipMemset(d_src, 0x3c, src_bytes));
    CK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_frames, frames.data(), frames.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    std::vector<float> gold((long)vblocks * 256), got((long)vblocks * 256);

    for (int lds_variant = 0; lds_variant < 3; ++lds_variant) {
        hipLaunchKernelGGL(victim, dim3(vblocks), dim3(256), 0, sv, d_table, d_frames, d_gold, rounds, lds_variant);
        CK(hipStreamSynchronize(sv));
        CK(hipMemcpy(gold.data(), d_gold, gold.size() * 4, hipMemcpyDeviceToHost));
        for (int mode = -1; mode < 3; ++mode) {   // -1: no aggressor
            long bad_values = 0, bad_trials = 0;
            long lane_hist[64] = {0};
            for (int t = 0; t < trials; ++t) {
                if (mode >= 0) hipLaunchKernelGGL(aggressor, dim3(2048), dim3(256), 65536, sa, d_src, d_sink, 400, mode, src_bytes);
                hipLaunchKernelGGL(victim, dim3(vblocks), dim3(256), 0, sv, d_table, d_frames, d_out, rounds, lds_variant);
                CK(hipStreamSynchronize(sv));
                CK(hipStreamSynchronize(sa));
                CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
                long nb = 0;
                for (long i = 0; i < (long)got.size(); ++i)
                    if (memcmp(&got[i], &gold[i], 4) != 0) {
                        if (bad_values + nb < 4) printf("   value %ld (block %ld thread %ld): %.9g -> %.9g\n", i, i / 256, i % 256, gold[i], got[i]);
                        ++nb; ++lane_hist[i & 63];
                    }
                bad_values += nb;
                bad_trials += nb > 0;
            }
            printf("victim %s | aggressor %s: %ld of %d trials differ, %ld values;", lds_variant == 0 ? "fp32, scattered LDS table" : lds_variant == 1 ? "fp32, no table reads" : "packed fp16",
                   mode < 0 ? "none" : mode == 0 ? "LDS-DMA" : mode == 1 ? "load+ds_write" : "MFMA", bad_trials, trials, bad_values);
            if (bad_values) {
                printf(" by lane quarter:");
                for (int q = 0; q < 4; ++q) { long s = 0; for (int l = 0; l < 16; ++l) s += lane_hist[q * 16 + l]; printf(" %ld", s); }
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
