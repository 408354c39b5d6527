// GEMV lab (not part of the product): the K = 4d decode GEMV (FC2 shape) alone, rotating over 32 weight matrices so every
// launch streams cold bytes; -DLAB_HOT_A / -DLAB_HOT_W make the activation / weight loads hit one cache-resident line set.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iwhisperx_mlx_amd/csrc tools/gemv_lab.hip -o tools/_bin/gemv_lab
#include "../whisperx_mlx_amd/csrc/skinny.hip"
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main() {
    const int M = 16, L = 32;
    struct Shape { const char* name; int N, K, tn, wide; } shapes[] = {{"fc2 tn8 w16", 1280, 5120, 8, 1}, {"fc2 tn16 w16", 1280, 5120, 16, 1},
                                                                        {"fc2 tn8 w8", 1280, 5120, 8, 0}, {"o-proj tn8", 1280, 1280, 8, 0},
                                                                        {"fc1-like no LN tn16", 5120, 1280, 16, 0}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        h16 *W, *A, *out, *bias;
        const size_t wel = (size_t)sh.N * sh.K;
        CK(hipMalloc(&W, wel * 2 * L)); CK(hipMemset(W, 0, wel * 2 * L));
        CK(hipMalloc(&A, (size_t)M * sh.K * 2)); CK(hipMemset(A, 0, (size_t)M * sh.K * 2));
        CK(hipMalloc(&out, (size_t)M * sh.N * 2)); CK(hipMalloc(&bias, sh.N * 2)); CK(hipMemset(bias, 0, sh.N * 2));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < L; ++l) {
                SkinnyArgs a{};
                a.A = A; a.lda = sh.K; a.W = W + (size_t)l * wel; a.ldw = sh.K; a.bias = bias; a.out_h = out; a.ldo = sh.N;
                a.M = M; a.N = sh.N; a.K = sh.K; a.tile_n = sh.tn; a.wide_block = sh.wide;
                CK(launch_skinny(a, s));
            }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s %.2f us per launch (graph of %d)\n", sh.name, ms * 1e3 / (4 * L), 4 * L);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
        CK(hipFree(W)); CK(hipFree(A)); CK(hipFree(out)); CK(hipFree(bias));
    }
    return 0;
}
