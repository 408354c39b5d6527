// Stand-alone harness around the product's own log-mel and GEMM launchers: runs the log-mel of
// 5 chunks alone, then again while another stream loops a GEMM of the encoder's FC2 shape, and
// reports how many fp32 log-mel values changed and in which mel bands.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwhisperx_mlx_amd/csrc -o tools/_bin/coresident_probe \
//         tools/coresident_probe.hip whisperx_mlx_amd/csrc/_obj/logmel.o <gemm object>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
#include "common.h"
#include "kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

int main(int argc, char** argv) {
    const int B = 5, NM = 80, NS = 480000, NF = 3000;
    const int gx = argc > 1 ? atoi(argv[1]) : 384, gy = argc > 2 ? atoi(argv[2]) : 7500, gk = argc > 3 ? atoi(argv[3]) : 1536;
    srand(3);
    std::vector<float> pcm((size_t)B * NS), filt((size_t)NM * 201, 0.f), tw(800), win(400);
    std::vector<int> lo(NM), len(NM), nv(B, NS);
    for (auto& v : pcm) v = ((float)rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    // a few harmonics so that the spectrum has loud and quiet bands
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < NS; ++i) pcm[(size_t)b * NS + i] = 0.3f * sinf(0.05f * i * (b + 1)) + 0.001f * pcm[(size_t)b * NS + i];
    for (int i = 0; i < 400; ++i) {
        tw[2 * i] = cosf(2.f * 3.14159265f * i / 400.f);
        tw[2 * i + 1] = -sinf(2.f * 3.14159265f * i / 400.f);
        win[i] = 0.5f - 0.5f * cosf(2.f * 3.14159265f * i / 400.f);
    }
    for (int m = 0; m < NM; ++m) {   // triangles of growing width, like the mel bank
        const int c = 2 + (int)(196.0 * pow((m + 1) / 81.0, 1.6)), w = 2 + m / 6;
        lo[m] = c - w < 0 ? 0 : c - w;
        const int hi = c + w > 200 ? 200 : c + w;
        len[m] = hi - lo[m] + 1;
        for (int k = lo[m]; k <= hi; ++k) filt[(size_t)m * 201 + k] = (1.f - fabsf((float)(k - c)) / (w + 1)) / (w + 1);
    }
    float *d_pcm, *d_filt, *d_tw, *d_win, *d_logspec, *d_out, *d_gold;
    int *d_lo, *d_len, *d_nv;
    unsigned* d_max;
    CK(hipMalloc(&d_pcm, pcm.size() * 4)); CK(hipMalloc(&d_filt, filt.size() * 4));
    CK(hipMalloc(&d_tw, 3200)); CK(hipMalloc(&d_win, 1600));
    CK(hipMalloc(&d_logspec, (size_t)B * NF * NM * 4)); CK(hipMalloc(&d_out, (size_t)B * NF * NM * 4));
    CK(hipMalloc(&d_gold, (size_t)B * NF * NM * 4));
    CK(hipMalloc(&d_lo, NM * 4)); CK(hipMalloc(&d_len, NM * 4)); CK(hipMalloc(&d_nv, B * 4)); CK(hipMalloc(&d_max, B * 4));
    CK(hipMemcpy(d_pcm, pcm.data(), pcm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_filt, filt.data(), filt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), 3200, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_win, win.data(), 1600, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_lo, lo.data(), NM * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, len.data(), NM * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nv, nv.data(), B * 4, hipMemcpyHostToDevice));

    // GEMM operands
    h16 *d_W, *d_A, *d_O, *d_bias;
    CK(hipMalloc(&d_W, (size_t)gx * gk * 2)); CK(hipMalloc(&d_A, (size_t)gy * gk * 2));
    CK(hipMalloc(&d_O, (size_t)gy * gx * 2)); CK(hipMalloc(&d_bias, (size_t)gx * 2));
    CK(hipMemset(d_W, 0x2c, (size_t)gx * gk * 2)); CK(hipMemset(d_A, 0x2c, (size_t)gy * gk * 2)); CK(hipMemset(d_bias, 0, (size_t)gx * 2));
    GemmArgs g{};
    g.X = d_W; g.ldx = gk; g.RX = gx; g.Y = d_A; g.ldy = gk; g.RY = gy; g.K = gk; g.bias = d_bias; g.out = d_O; g.ldo = gx;

    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    LogmelArgs la{d_pcm, (long)NS, d_nv, d_filt, d_lo, d_len, d_tw, d_win, d_logspec, d_max, B, NM};
    auto logmel = [&](float* out) {
        CK(launch_logmel(la, sv));
        CK(launch_logmel_finalize(d_logspec, d_max, out, nullptr, 0, 0, B, NM, sv));
    };
    logmel(d_gold);
    CK(hipStreamSynchronize(sv));
    const size_t n = (size_t)B * NF * NM;
    std::vector<float> gold(n), got(n);
    CK(hipMemcpy(gold.data(), d_gold, n * 4, hipMemcpyDeviceToHost));
    for (int with = 0; with < 2; ++with) {
        long bad_trials = 0, bad = 0;
        std::vector<long> band(NM, 0);
        for (int t = 0; t < 10; ++t) {
            if (with) for (int r = 0; r < 40; ++r) CK(launch_gemm_f16(g, 1, false, sa));
            logmel(d_out);
            CK(hipStreamSynchronize(sv));
            CK(hipStreamSynchronize(sa));
            CK(hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost));
            long nb = 0;
            for (size_t i = 0; i < n; ++i) if (memcmp(&got[i], &gold[i], 4)) { ++nb; ++band[i % NM]; }
            bad += nb; bad_trials += nb > 0;
        }
        printf("log-mel %s: %ld of 10 trials differ, %ld values", with ? "beside the GEMM" : "alone", bad_trials, bad);
        if (bad) { printf("; bands:"); for (int m = 0; m < NM; ++m) if (band[m]) printf(" %d", m); }
        printf("\n");
    }
    return 0;
}
