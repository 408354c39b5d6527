// Kernel lab for the dense GEMM (not part of the product): builds gemm.hip into a standalone
// binary, times shapes with HIP events on random data, spot-checks results against fp64 host dot
// products and (with -DWX_GEMM_LAB stamps) dumps s_memtime stamps of block 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWX_GEMM_LAB -Iwhisperx_mlx_amd/csrc tools/gemm_lab.hip -o gpurun_out/gemm_lab
#include "../whisperx_mlx_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    struct Shape { const char* name; int M, N, K; bool gelu; };
    std::vector<Shape> shapes = {{"fc1", 24000, 5120, 1280, true}, {"f1ng", 24000, 5120, 1280, false}, {"qkv", 24000, 3840, 1280, false}, {"fc2", 24000, 1280, 5120, false},
                                 {"proj", 24000, 1280, 1280, false}, {"sq4096", 4096, 4096, 4096, false},
                                 {"sq8192", 8192, 8192, 8192, false}};
    const int iters = 10;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    for (const Shape& sh : shapes) {
        if (argc > 1 && !strstr(argv[1], sh.name)) continue;
        const size_t nx = (size_t)sh.N * sh.K, ny = (size_t)sh.M * sh.K, no = (size_t)sh.M * sh.N;
        std::vector<h16> hx(nx), hy(ny), ho(no), hb(sh.N);
        for (auto& v : hx) v = (h16)(U(rng) * 0.05f);
        for (auto& v : hy) v = (h16)U(rng);
        for (auto& v : hb) v = (h16)U(rng);
        h16 *dx, *dy, *dout, *db;
        CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&dout, no * 2)); CK(hipMalloc(&db, sh.N * 2));
        CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), sh.N * 2, hipMemcpyHostToDevice));
        GemmArgs a{};
        a.X = dx; a.ldx = sh.K; a.RX = sh.N;
        a.Y = dy; a.ldy = sh.K; a.RY = sh.M;
        a.K = sh.K; a.bias = db; a.out = dout; a.ldo = sh.N;
#ifdef WX_GEMM_LAB
        long long* dst; CK(hipMalloc(&dst, 8 * 4096 * 8)); CK(hipMemset(dst, 0, 8 * 4096 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(wx_lab_stamps), &dst, sizeof(dst)));
#endif
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(launch_gemm_f16(a, 1, sh.gelu, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) CK(launch_gemm_f16(a, 1, sh.gelu, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, tf = 2.0 * sh.M * sh.N * sh.K / us * 1e-6;
        CK(hipMemcpy(ho.data(), dout, no * 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int s = 0; s < 256; ++s) {
            const int m = (s < 8) ? sh.M - 1 - s : rng() % sh.M, n = (s < 8) ? sh.N - 1 - s * 3 : rng() % sh.N;
            double r = (double)(float)hb[n];
            for (int k = 0; k < sh.K; ++k) r += (double)(float)hx[(size_t)n * sh.K + k] * (double)(float)hy[(size_t)m * sh.K + k];
            if (sh.gelu) r = 0.5 * r * (1 + erf(r / sqrt(2.0)));
            const double d = fabs(r - (double)(float)ho[(size_t)m * sh.N + n]) / (1.0 + fabs(r));
            if (d > maxerr) maxerr = d;
        }
        printf("%-8s M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s  max rel err %.2e\n", sh.name, sh.M, sh.N, sh.K, us, tf, maxerr);
#ifdef WX_GEMM_LAB
        std::vector<long long> st(8 * 4096);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        for (int w : {0, 4}) {
            printf("  wave %d stamp deltas:", w);
            for (int i = 1; i < 4096 && st[w * 4096 + i]; ++i) {
                if (i > 100) break;
                printf(" %lld", st[w * 4096 + i] - st[w * 4096 + i - 1]);
            }
            printf("\n");
        }
        CK(hipFree(dst));
#endif
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dy)); CK(hipFree(dout)); CK(hipFree(db));
    }
    return 0;
}
